// Radix-2 NTT over Fr for gfx950 and the fused computeH pipeline.
// Replaces fft.Domain.FFT / FFTInverse (gnark-crypto ecc/bn254/fr/fft) and computeH (gnark
// backend/groth16/bn254/prove.go) on the path reached from /root/reference/mt.go:496
// (SURVEY.md 8a rows a3/a4).  Pass structure and index arithmetic: ntt_tile.cuh.
//
// HBM layout: the caller's array of 32-byte Montgomery elements, transformed in place.  Each pass
// is one launch; a workgroup owns one LDS tile (<= 2^log_e elements, two 16-byte planes), loads
// it with >= 256-byte contiguous runs, runs log2(R) butterfly stages out of LDS and stores it
// back.  Root tables: one 1024-entry table serves every in-tile stage; inter-pass twiddles and
// coset powers come from two sqrt(N)-sized tables and one product (no N-sized table is ever read).
// Bound: HBM for the load/store (64 B per element per pass), VALU (one 256-bit Montgomery product
// per butterfly) for the stages -- see DESIGN.md for the measured split.
#include "ctx.h"
#include "ntt_wave.cuh"
#include "msm2_core.cuh"   // the Z MSM's digit count rides in computeH's last launch (mi_ctx::zhook)
#include <cstring>
#include <cstdlib>

struct NttState {
    u32 log_n = 0xffffffffu;
    Fr *small_f = nullptr, *small_i = nullptr, *tw64k_f = nullptr, *tw64k_i = nullptr;   // N independent
    Fr *tw_lo_f = nullptr, *tw_hi_f = nullptr, *tw_lo_i = nullptr, *tw_hi_i = nullptr;
    Fr *g_lo = nullptr, *g_hi = nullptr, *gi_lo = nullptr, *gi_hi = nullptr, *ninv = nullptr;
    Fr *g_hi_n = nullptr, *gi_hi_nd = nullptr;   // computeH-internal: g^(j 2^h) / N  and  g^-(j 2^h) / N * den
    Fr *ninv_den = nullptr;                      // computeH-internal: den / N
    u32 tw_h = 0;
    // computeH's direct factor tables in the data's layout (NttPass::tw_direct / sc_direct), N entries each, built for one
    // (log_n, first radix): twiddles of the M = N pass for the inverse and the forward root, coset shifts g^bitrev(i) / N and
    // g^-bitrev(i) * den / N.  4 x 32 B x N of HBM (1 GiB at N = 2^23) buy 11 of computeH's ~107 products per element.
    Fr *d_tw_inv = nullptr, *d_tw_fwd = nullptr, *d_sc_fwd = nullptr, *d_sc_inv = nullptr;
    u32 d_log_n = 0xffffffffu, d_log_r0 = 0, d_npass = 0;
    u32 direct_min_log_n = 12;   // below this the tables are not worth a launch
    // plan knobs (mi_debug_set_ntt_plan)
    // defaults from the tools/tune.py sweep at N = 2^23: 2^9-element tiles (16 KiB of LDS, 8 workgroups of 256
    // threads per CU), radices 2^7 * 2^7 * 2^9: 1.46 ms per transform vs 1.99 ms for 2^11 tiles
    u32 log_e = 9, max_contig = 9, max_strided = 7, threads = 256;
    u32 lds_floor = 0;     // bytes of LDS every pass workgroup requests at least: caps the workgroups per CU (0 = only what the tile needs)
    u32 plan_set = 0;      // the knobs above were set by the caller: no per-size defaults
    u32 wave_stages = 1;   // passes of radix >= 2^7 run their last / first seven stages in registers (k_ntt_pass_wave); 0 = all through LDS
    u32 fuse_pair = 1;     // computeH: inverse last pass + coset first pass of a and b as one launch (k_ntt_contig_pair)
    u32 fuse_triple = 1;   // computeH: coset last pass of a and of b + the product a b + the last transform's first pass as one launch (k_ntt_strided_triple)
    u32 fuse_last = 1;     // computeH: the last pass of den FFTInverse(c) + the last transform's last pass (which subtracts it) as one launch (k_ntt_contig_last_sub)
};
// Tile / radix knobs for a transform of 2^log_n: the caller's (mi_debug_set_ntt_plan) or, untouched, the measured best per size
// (tools/ntt_probe.py sweep, profiles/r02_tune_ntt_sweep.json): 2^9 tiles and radices 2^7 2^7 2^9 up to 2^23; 2^10 tiles and
// 2^8 2^8 2^8 at 2^24 (16.2 vs 17.4 ms per computeH); 2^10 tiles and 2^8 2^8(2^7) 2^10 from 2^25 on (71.9 vs 74.0 ms at 2^26).
struct NttKnobs { u32 log_e, max_contig, max_strided; };
static NttKnobs knobs_for(const struct NttState *st, u32 log_n);
static NttState *state_of(mi_ctx *ctx) {
    static_assert(sizeof(NttState) <= 384, "NttState lives in ctx->ntt_state");
    return reinterpret_cast<NttState *>(ctx->ntt_state);
}

__global__ void k_ntt_pass(Fr *dst, const Fr *src, NttPass p, NttTables t);
__global__ void k_ntt_pass_wave(Fr *dst, const Fr *src, NttPass p, NttTables t);
__global__ void k_ntt_contig_pair(Fr *data, NttPass pi, NttTables ti, NttPass pf, NttTables tf);
__global__ void k_ntt_strided_triple(Fr *a, const Fr *b, NttPass pc, NttTables tc, NttPass pl, NttTables tl);
__global__ void k_ntt_strided_triple8(Fr *a, const Fr *b, NttPass pc, NttTables tc, NttPass pl, NttTables tl);
__global__ void k_ntt_contig_last_sub(Fr *A, const Fr *Cin, NttPass pa, NttTables ta, NttPass pc, NttTables tc);
template <int CC> __global__ void k_ntt_contig_last_sub_count(Fr *A, const Fr *Cin, NttPass pa, NttTables ta, NttPass pc, NttTables tc, Msm2Shape zs, u32 *C1);
void mi_ntt_state_init(mi_ctx *ctx) {
    static_assert(sizeof(NttState) <= sizeof(ctx->ntt_state), "NttState lives in ctx->ntt_state");
    new (ctx->ntt_state) NttState();
    (void)hipFuncSetAttribute((const void *)k_ntt_pass, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)k_ntt_pass_wave, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)k_ntt_contig_pair, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)k_ntt_strided_triple, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)k_ntt_strided_triple8, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)k_ntt_contig_last_sub, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)k_ntt_contig_last_sub_count<17>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)k_ntt_contig_last_sub_count<18>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)k_ntt_contig_last_sub_count<19>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)k_ntt_contig_last_sub_count<20>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)k_ntt_contig_last_sub_count<21>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)k_ntt_contig_last_sub_count<22>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}
bool mi_ntt_set_knob(mi_ctx *ctx, const char *name, int64_t value) {
    if (!std::strcmp(name, "ntt_lds_floor_kb") && value >= 0 && value <= 160) { state_of(ctx)->lds_floor = (u32)value * 1024u; return true; }
    return false;
}
void mi_ntt_state_free(mi_ctx *ctx) {
    NttState *st = state_of(ctx);
    Fr **all[] = {&st->small_f, &st->small_i, &st->tw64k_f, &st->tw64k_i, &st->g_hi_n, &st->gi_hi_nd, &st->tw_lo_f, &st->tw_hi_f, &st->tw_lo_i, &st->tw_hi_i,
                  &st->g_lo, &st->g_hi, &st->gi_lo, &st->gi_hi, &st->ninv, &st->ninv_den, &st->d_tw_inv, &st->d_tw_fwd, &st->d_sc_fwd, &st->d_sc_inv};
    for (Fr **p : all) if (*p) { (void)hipFree(*p); *p = nullptr; }
}

// mi_ctx_trim: every table goes; the markers say "nothing built", the plan knobs stay
void mi_ntt_state_trim(mi_ctx *ctx) {
    mi_ntt_state_free(ctx);
    NttState *st = state_of(ctx);
    st->log_n = 0xffffffffu; st->d_log_n = 0xffffffffu; st->d_log_r0 = 0; st->d_npass = 0;
}
// bytes of device memory the context's NTT tables take (mi_get_mem_ledger)
size_t mi_ntt_table_bytes(mi_ctx *ctx) {
    const NttState *st = state_of(ctx);
    size_t b = 0;
    if (st->small_f) b += (size_t)(2 * 2048 + 2 * 65536) * sizeof(Fr);
    if (st->log_n != 0xffffffffu) {
        const size_t nlo = (size_t)1 << st->tw_h, nhi = (size_t)1 << (st->log_n - st->tw_h);
        b += (4 * nlo + 6 * nhi + 2) * sizeof(Fr);
    }
    for (const Fr *p : {st->d_tw_inv, st->d_tw_fwd, st->d_sc_fwd, st->d_sc_inv}) if (p) b += sizeof(Fr) << st->d_log_n;
    return b;
}

static NttKnobs knobs_for(const NttState *st, u32 log_n) {
    if (st->plan_set) return NttKnobs{st->log_e, st->max_contig, st->max_strided};
    if (log_n == 24) return NttKnobs{10, 8, 8};
    if (log_n >= 25) return NttKnobs{10, 10, 8};
    return NttKnobs{st->log_e, st->max_contig, st->max_strided};
}

// out[j] = c * base^(j << shift)
__global__ void k_pow_table(Fr *out, u32 count, Fr base, Fr c, u32 shift) {
    u32 j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count) return;
    u64 e = (u64)j << shift;
    Fr acc = c, b = base;
    while (e) {
        if (e & 1) acc = acc * b;
        b = fe_sqr(b);
        e >>= 1;
    }
    out[j] = acc;
}

__global__ void __launch_bounds__(1024) k_ntt_pass(Fr *dst, const Fr *src, NttPass p, NttTables t) {
    extern __shared__ U4 lds[];
    const u64 tile = blockIdx.x;
    ntt_tile_load(p, t, src, tile, threadIdx.x, blockDim.x, lds);
    __syncthreads();
    for (u32 s = 0; s < p.log_r; s++) {
        ntt_tile_stage(p, t, s, threadIdx.x, blockDim.x, lds);
        __syncthreads();
    }
    ntt_tile_store(p, t, dst, tile, threadIdx.x, blockDim.x, lds);
}

// The same pass with the seven stages at distance <= 64 of every 128-row group done in REGISTERS by one wavefront
// (ntt_wave.cuh): the tile makes two LDS round trips (in and out of the registers) plus one per stage at distance >= 128,
// instead of one per stage, and two barriers plus one per such stage.  Needs log_r >= 7.
#ifndef MI_NTT_WAVES_PER_EU
#define MI_NTT_WAVES_PER_EU 1   // experiment switch: 6 = no change (79 VGPRs either way), 8 = 64 VGPRs + 52 B of scratch, slower alone and in a proof
#endif
__global__ void __launch_bounds__(256, MI_NTT_WAVES_PER_EU) k_ntt_pass_wave(Fr *dst, const Fr *src, NttPass p, NttTables t) {
    extern __shared__ U4 lds[];
    const u64 tile = blockIdx.x;
    const u32 PL = ntt_plane_slots(p);
    ntt_tile_load(p, t, src, tile, threadIdx.x, blockDim.x, lds);
    __syncthreads();
    if (!p.dit)   // DIF: distances R/2 ... 128 first
        for (u32 s = 0; s + 7 < p.log_r; s++) {
            ntt_tile_stage(p, t, s, threadIdx.x, blockDim.x, lds);
            __syncthreads();
        }
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const u32 nsub = 1u << (p.log_r + p.log_c - 7);
    for (u32 sb = wave; sb < nsub; sb += nwaves) {   // sub-block = 128 consecutive rows of one column: touched by this wave only
        const u32 col = sb & ((1u << p.log_c) - 1), base = (sb >> p.log_c) << 7;
        u32 ra, rb, wa, wb;
        if (!p.dit) { ra = base + lane; rb = ra + 64; wa = base + 2 * lane; wb = wa + 1; }
        else { ra = base + 2 * lane; rb = ra + 1; wa = base + lane; wb = wa + 64; }
        Fr x0 = lds_get(lds, PL, ntt_lds_slot(p, ra, col)), x1 = lds_get(lds, PL, ntt_lds_slot(p, rb, col));
        wave_ntt128(x0, x1, lane, p.dit != 0, t.small);
        lds_put(lds, PL, ntt_lds_slot(p, wa, col), x0);
        lds_put(lds, PL, ntt_lds_slot(p, wb, col), x1);
    }
    __syncthreads();
    if (p.dit)    // DIT: distances 128 ... R/2 last
        for (u32 s = 7; s < p.log_r; s++) {
            ntt_tile_stage(p, t, s, threadIdx.x, blockDim.x, lds);
            __syncthreads();
        }
    ntt_tile_store(p, t, dst, tile, threadIdx.x, blockDim.x, lds);
}

// computeH's a and b: the LAST pass of the inverse transform (DIF, contiguous tiles) and the FIRST pass of the coset transform that
// follows (DIT, the same contiguous tiles, coset shift on the way in) as one launch -- the tile is loaded once, runs the inverse
// stages, is multiplied by the shift, runs the forward stages and is stored once; in the register stages the two transforms meet
// without an LDS round trip (a lane ends the DIF stages holding positions 2L, 2L + 1 -- what the DIT stages start from).  One store
// and one load of the vector (0.54 GB at N = 2^23) and one launch less per vector.  pi / ti: the inverse pass as ntt_run would have
// launched it, pf / tf: the forward one.  Radix >= 2^7, equal tile shapes.
__global__ void __launch_bounds__(256) k_ntt_contig_pair(Fr *data, NttPass pi, NttTables ti, NttPass pf, NttTables tf) {
    extern __shared__ U4 lds[];
    const u64 tile = blockIdx.x;
    const u32 PL = ntt_plane_slots(pi);
    ntt_tile_load(pi, ti, data, tile, threadIdx.x, blockDim.x, lds);
    __syncthreads();
    for (u32 s = 0; s + 7 < pi.log_r; s++) {   // DIF distances R/2 ... 128
        ntt_tile_stage(pi, ti, s, threadIdx.x, blockDim.x, lds);
        __syncthreads();
    }
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const u32 nsub = 1u << (pi.log_r + pi.log_c - 7);
    for (u32 sb = wave; sb < nsub; sb += nwaves) {
        const u32 col = sb & ((1u << pi.log_c) - 1), base = (sb >> pi.log_c) << 7;
        Fr x0 = lds_get(lds, PL, ntt_lds_slot(pi, base + lane, col)), x1 = lds_get(lds, PL, ntt_lds_slot(pi, base + lane + 64, col));
        wave_ntt128(x0, x1, lane, false, ti.small);
        // what the inverse pass would have stored at rows 2L, 2L + 1 and the forward pass loaded with its edge factor
        const u32 r0 = base + 2 * lane, r1 = r0 + 1;
        Fr f;
        if (ntt_edge_factor(pi, ti, r0, ntt_global_index(pi, tile, r0, col), 1, f)) x0 = fe_mul_lazy(x0, f);
        if (ntt_edge_factor(pf, tf, r0, ntt_global_index(pf, tile, r0, col), 0, f)) x0 = fe_mul_lazy(x0, f);
        if (ntt_edge_factor(pi, ti, r1, ntt_global_index(pi, tile, r1, col), 1, f)) x1 = fe_mul_lazy(x1, f);
        if (ntt_edge_factor(pf, tf, r1, ntt_global_index(pf, tile, r1, col), 0, f)) x1 = fe_mul_lazy(x1, f);
        wave_ntt128(x0, x1, lane, true, tf.small);
        lds_put(lds, PL, ntt_lds_slot(pf, base + lane, col), x0);
        lds_put(lds, PL, ntt_lds_slot(pf, base + lane + 64, col), x1);
    }
    __syncthreads();
    for (u32 s = 7; s < pf.log_r; s++) {   // DIT distances 128 ... R/2
        ntt_tile_stage(pf, tf, s, threadIdx.x, blockDim.x, lds);
        __syncthreads();
    }
    ntt_tile_store(pf, tf, data, tile, threadIdx.x, blockDim.x, lds);
}

// computeH's OTHER seam: the LAST pass of the coset FFT of a and of b (DIT, the strided M = N pass), the pointwise product a b, and the
// FIRST pass of the last transform (DIF, inverse on the coset, the same strided tiles) as one launch.  A workgroup loads a's tile
// (DIT pre-twiddle on the way in), runs the seven register stages and keeps the result in registers (a lane ends DIT holding rows L,
// L + 64 -- exactly what the DIF stages start from), loads b's tile into the same LDS, does the same, multiplies, runs the DIF stages
// and stores through LDS with the DIF post-twiddle.  Per element: the same products; four passes of a vector through HBM (store a,
// store b, load a, load b: 1.07 GB at N = 2^23) and two launches less.  pc / tc: the coset FFT's last pass as ntt_run would have
// launched it (for a and for b alike), pl / tl: the last transform's first pass (without load_mul).  Radix 2^7 exactly (no stage
// through LDS) and one 128-row sub-block per wave (blockDim = 64 x sub-blocks of the tile).
__global__ void __launch_bounds__(512) k_ntt_strided_triple(Fr *a, const Fr *b, NttPass pc, NttTables tc, NttPass pl, NttTables tl) {
    extern __shared__ U4 lds[];
    const u64 tile = blockIdx.x;
    const u32 PL = ntt_plane_slots(pc);
    const u32 lane = threadIdx.x & 63, sb = threadIdx.x >> 6;
    const u32 col = sb & ((1u << pc.log_c) - 1), base = (sb >> pc.log_c) << 7;
    const u32 s0 = ntt_lds_slot(pc, base + 2 * lane, col), s1 = ntt_lds_slot(pc, base + 2 * lane + 1, col);
    ntt_tile_load(pc, tc, a, tile, threadIdx.x, blockDim.x, lds);
    __syncthreads();
    Fr a0 = lds_get(lds, PL, s0), a1 = lds_get(lds, PL, s1);
    wave_ntt128(a0, a1, lane, true, tc.small);     // -> rows L, L + 64
    __syncthreads();                               // every wave has read a's tile
    ntt_tile_load(pc, tc, b, tile, threadIdx.x, blockDim.x, lds);
    __syncthreads();
    Fr x0 = lds_get(lds, PL, s0), x1 = lds_get(lds, PL, s1);
    wave_ntt128(x0, x1, lane, true, tc.small);
    x0 = ntt_mul_lazy2(x0, a0); x1 = ntt_mul_lazy2(x1, a1);   // DIT left both factors below 4p; the DIF stages want [0, 2p)
    wave_ntt128(x0, x1, lane, false, tl.small);    // rows L, L + 64 -> positions 2L, 2L + 1
    __syncthreads();                               // every wave has read b's tile
    lds_put(lds, PL, ntt_lds_slot(pl, base + 2 * lane, col), x0);
    lds_put(lds, PL, ntt_lds_slot(pl, base + 2 * lane + 1, col), x1);
    __syncthreads();
    ntt_tile_store(pl, tl, a, tile, threadIdx.x, blockDim.x, lds);
}
// The same seam for a first radix of 2^8 (the plans of N = 2^24 and 2^26): the DIT pass ends, and the DIF pass begins, with the stage at
// distance 128 -- on the SAME pairs (r, r + 128).  Each 128-row half of a column goes through the register stages of one wave and back
// to LDS; a thread then takes its pairs, does the DIT butterfly and keeps the two results in registers (for a), does the same for b,
// multiplies, does the DIF butterfly on the products and writes them to LDS for the DIF register stages.  256 threads per 256-row x
// 2^log_c-column tile: 2^log_c butterflies per thread at distance 128 (2^(log_c + 1) elements of a held in registers: log_c <= 2).
__global__ void __launch_bounds__(256) k_ntt_strided_triple8(Fr *a, const Fr *b, NttPass pc, NttTables tc, NttPass pl, NttTables tl) {
    extern __shared__ U4 lds[];
    const u64 tile = blockIdx.x;
    const u32 PL = ntt_plane_slots(pc);
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const u32 C = 1u << pc.log_c, nsub = 2u << pc.log_c;   // 128-row sub-blocks of the tile
    const u32 nb = C >> 1;                                  // butterflies at distance 128 per thread: 128 * C over 256 threads
    Fr keep[4];                                             // a's results of this thread's pairs (nb <= 2 pairs)
    auto reg_stages = [&](const NttPass &p, bool dit, const Fr *small) {
        for (u32 sbk = wave; sbk < nsub; sbk += nwaves) {
            const u32 col = sbk & (C - 1), base = (sbk >> p.log_c) << 7;
            u32 ra, rb, wa, wb;
            if (!dit) { ra = base + lane; rb = ra + 64; wa = base + 2 * lane; wb = wa + 1; }
            else { ra = base + 2 * lane; rb = ra + 1; wa = base + lane; wb = wa + 64; }
            Fr x0 = lds_get(lds, PL, ntt_lds_slot(p, ra, col)), x1 = lds_get(lds, PL, ntt_lds_slot(p, rb, col));
            wave_ntt128(x0, x1, lane, dit, small);
            lds_put(lds, PL, ntt_lds_slot(p, wa, col), x0);
            lds_put(lds, PL, ntt_lds_slot(p, wb, col), x1);
        }
    };
    // pair k of this thread: butterfly index bf = threadIdx.x + 256 k over (column, row j < 128), rows j and j + 128
    ntt_tile_load(pc, tc, a, tile, threadIdx.x, blockDim.x, lds);
    __syncthreads();
    reg_stages(pc, true, tc.small);
    __syncthreads();
#pragma unroll
    for (u32 k = 0; k < 2; k++) {   // (constant trip count: keep[] stays in registers)
        if (k >= nb) break;
        const u32 bf = threadIdx.x + 256 * k, col = bf & (C - 1), j = bf >> pc.log_c;
        Fr x = lds_get(lds, PL, ntt_lds_slot(pc, j, col));
        Fr y = lds_get(lds, PL, ntt_lds_slot(pc, j + 128, col));
        ntt_bfly_dit(x, y, j ? &tc.small[j << 4] : nullptr);           // w_256^j = w_4096^(16 j)
        keep[2 * k] = x; keep[2 * k + 1] = y;
    }
    __syncthreads();
    ntt_tile_load(pc, tc, b, tile, threadIdx.x, blockDim.x, lds);
    __syncthreads();
    reg_stages(pc, true, tc.small);
    __syncthreads();
#pragma unroll
    for (u32 k = 0; k < 2; k++) {
        if (k >= nb) break;
        const u32 bf = threadIdx.x + 256 * k, col = bf & (C - 1), j = bf >> pc.log_c;
        Fr x = lds_get(lds, PL, ntt_lds_slot(pc, j, col));
        Fr y = lds_get(lds, PL, ntt_lds_slot(pc, j + 128, col));
        ntt_bfly_dit(x, y, j ? &tc.small[j << 4] : nullptr);
        Fr p0 = ntt_mul_lazy2(x, keep[2 * k]), p1 = ntt_mul_lazy2(y, keep[2 * k + 1]);   // the products at rows j, j + 128, below 2p
        ntt_bfly_dif(p0, p1, j ? &tl.small[j << 4] : nullptr);                           // DIF butterfly at distance 128
        lds_put(lds, PL, ntt_lds_slot(pl, j, col), p0);
        lds_put(lds, PL, ntt_lds_slot(pl, j + 128, col), p1);
    }
    __syncthreads();
    reg_stages(pl, false, tl.small);
    __syncthreads();
    ntt_tile_store(pl, tl, a, tile, threadIdx.x, blockDim.x, lds);
}

// computeH's LAST seam: the last (contiguous, DIF) pass of den FFTInverse(c) and the last pass of the last transform work on the same
// contiguous tiles, and the second subtracts the first's output element by element.  One launch: c's tile goes through its stages and its
// den / N scaling and stays in REGISTERS (a thread keeps the elements it would have stored), the main tile follows through the same LDS,
// and the store subtracts, canonicalises and writes h.  c's last store and the store_sub read (0.54 GB at N = 2^23) and one launch go.
// pa / ta: the last transform's last pass as ntt_run would have launched it (store_sub unset), pc / tc: c's.  Equal tile shapes, radix >= 2^7,
// at most four elements per thread.
__device__ __forceinline__ void ntt_wave_pass_to_lds(const NttPass &p, const NttTables &t, const Fr *src, u64 tile, U4 *lds) {
    const u32 PL = ntt_plane_slots(p);
    ntt_tile_load(p, t, src, tile, threadIdx.x, blockDim.x, lds);
    __syncthreads();
    for (u32 s = 0; s + 7 < p.log_r; s++) {   // DIF distances R/2 ... 128
        ntt_tile_stage(p, t, s, threadIdx.x, blockDim.x, lds);
        __syncthreads();
    }
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const u32 nsub = 1u << (p.log_r + p.log_c - 7);
    for (u32 sb = wave; sb < nsub; sb += nwaves) {
        const u32 col = sb & ((1u << p.log_c) - 1), base = (sb >> p.log_c) << 7;
        Fr x0 = lds_get(lds, PL, ntt_lds_slot(p, base + lane, col)), x1 = lds_get(lds, PL, ntt_lds_slot(p, base + lane + 64, col));
        wave_ntt128(x0, x1, lane, false, t.small);
        lds_put(lds, PL, ntt_lds_slot(p, base + 2 * lane, col), x0);
        lds_put(lds, PL, ntt_lds_slot(p, base + 2 * lane + 1, col), x1);
    }
    __syncthreads();
}
// CC >= 17: the same launch also COUNTS the digits of every h coefficient it stores, for the fixed-base sort of the Z MSM that follows
// (msm2_core.cuh: a contiguous tile of E elements is E / MSM2_SLICE whole slices of that sort; counters in LDS behind the tile's two planes,
// written to C1[group][slice] exactly as k_msm2_count would have).  CC = -1: no count (the kernel every other caller gets).
template <int CC>
__device__ __forceinline__ void ntt_contig_last_sub_body(Fr *A, const Fr *Cin, const NttPass &pa, const NttTables &ta, const NttPass &pc, const NttTables &tc,
                                                         const Msm2Shape &zs, u32 *C1) {
    extern __shared__ U4 lds[];
    const u64 tile = blockIdx.x;
    const u32 E = 1u << (pa.log_r + pa.log_c), PL = ntt_plane_slots(pa);
    u32 *cnt = reinterpret_cast<u32 *>(lds + 2 * PL);
    const u32 nsl = E / MSM2_SLICE;
    if (CC >= 0) for (u32 k = threadIdx.x; k < nsl * zs.ngroups; k += blockDim.x) cnt[k] = 0;   // (barriers follow before the first count)
    Fr keep[4];
    ntt_wave_pass_to_lds(pc, tc, Cin, tile, lds);
#pragma unroll
    for (u32 k = 0; k < 4; k++) {   // (constant trip count: keep[] stays in registers)
        const u32 e = threadIdx.x + k * blockDim.x;
        if (e >= E) break;
        const u32 rho = e & ((1u << pc.log_r) - 1), col = e >> pc.log_r;   // contiguous pass: the order that is contiguous in global memory
        Fr v = lds_get(lds, PL, ntt_lds_slot(pc, rho, col)), f;
        if (ntt_edge_factor(pc, tc, rho, ntt_global_index(pc, tile, rho, col), 1, f)) v = fe_mul_lazy(v, f);
        keep[k] = v;
    }
    __syncthreads();   // every thread has read c's tile
    ntt_wave_pass_to_lds(pa, ta, A, tile, lds);
#pragma unroll
    for (u32 k = 0; k < 4; k++) {
        const u32 e = threadIdx.x + k * blockDim.x;
        if (e >= E) break;
        const u32 rho = e & ((1u << pa.log_r) - 1), col = e >> pa.log_r;
        const u64 g = ntt_global_index(pa, tile, rho, col);
        Fr v = lds_get(lds, PL, ntt_lds_slot(pa, rho, col)), f;
        if (ntt_edge_factor(pa, ta, rho, g, 1, f)) v = fe_mul_lazy(v, f);
        const Fr hv = fe_canon(fe_sub_plus2p(fe_condsub_2p(v), fe_condsub_2p(keep[k])));
        A[g] = hv;
        if (CC >= 0 && g < zs.n) {
            Msm2Digits dg;
            dg.start(hv, true);
            msm2_count_one<(CC >= 0 ? (u32)CC : 0u)>(zs, dg, cnt + (e / MSM2_SLICE) * zs.ngroups);
        }
    }
    if (CC >= 0) {
        __syncthreads();
        const u32 s0 = (u32)tile * nsl;
        for (u32 k = threadIdx.x; k < nsl * zs.ngroups; k += blockDim.x) {
            const u32 j = k / zs.ngroups, gq = k % zs.ngroups;
            if (s0 + j < zs.nslices) C1[(size_t)gq * zs.nslices + s0 + j] = cnt[k];
        }
    }
}
__global__ void __launch_bounds__(256) k_ntt_contig_last_sub(Fr *A, const Fr *Cin, NttPass pa, NttTables ta, NttPass pc, NttTables tc) {
    ntt_contig_last_sub_body<-1>(A, Cin, pa, ta, pc, tc, Msm2Shape{}, nullptr);
}
template <int CC>
__global__ void __launch_bounds__(256) k_ntt_contig_last_sub_count(Fr *A, const Fr *Cin, NttPass pa, NttTables ta, NttPass pc, NttTables tc, Msm2Shape zs, u32 *C1) {
    ntt_contig_last_sub_body<CC>(A, Cin, pa, ta, pc, tc, zs, C1);
}

// direct factor tables (NttPass::tw_direct / sc_direct), one thread per entry, square-and-multiply
__global__ void k_tw_layout(Fr *out, u32 log_n, u32 log_r, Fr w) {   // the M = N pass: element g = (rho << log_s) | lo
    const u32 g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (1u << log_n)) return;
    const u32 log_s = log_n - log_r;
    u32 e = (g & ((1u << log_s) - 1)) * bitrev_u32(g >> log_s, log_r);   // < N
    Fr acc = Fr::one(), b = w;
    while (e) { if (e & 1) acc = acc * b; b = fe_sqr(b); e >>= 1; }
    out[g] = acc;
}
__global__ void k_sc_layout(Fr *out, u32 log_n, Fr base, Fr c) {     // out[i] = c * base^bitrev_N(i)
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (1u << log_n)) return;
    u32 e = bitrev_u32(i, log_n);
    Fr acc = c, b = base;
    while (e) { if (e & 1) acc = acc * b; b = fe_sqr(b); e >>= 1; }
    out[i] = acc;
}

static Fr host_fr_from_u64x4(u64 a, u64 b, u64 c, u64 d) {
    Fr t;
    t.l[0] = (u32)a; t.l[1] = (u32)(a >> 32); t.l[2] = (u32)b; t.l[3] = (u32)(b >> 32);
    t.l[4] = (u32)c; t.l[5] = (u32)(c >> 32); t.l[6] = (u32)d; t.l[7] = (u32)(d >> 32);
    return fe_to_mont(t);
}
static Fr domain_generator(u32 log_n) {  // fft.NewDomain: Generator = root^(2^(28-log_n))
    Fr g = host_fr_from_u64x4(0x9bd61b6e725b19f0ull, 0x402d111e41112ed4ull, 0x00e0a7eb8ef62abcull, 0x2a3c09f0a58a7e85ull);
    for (u32 k = log_n; k < 28; k++) g = fe_sqr(g);
    return g;
}

static int32_t build_table(mi_ctx *ctx, Fr **slot, u32 count, const Fr &base, const Fr &c, u32 shift) {
    if (*slot) { MI_CHECK_HIP(ctx, hipFree(*slot)); *slot = nullptr; }
    MI_CHECK_HIP(ctx, hipMalloc((void **)slot, sizeof(Fr) * (count ? count : 1)));
    hipLaunchKernelGGL(k_pow_table, dim3((count + 127) / 128), dim3(128), 0, ctx->stream, *slot, count, base, c, shift);
    MI_CHECK_HIP(ctx, hipGetLastError());
    return MI_OK;
}

static int32_t ensure_tables(mi_ctx *ctx, u32 log_n) {
    NttState *st = state_of(ctx);
    if (!st->small_f) {
        Fr w4096 = domain_generator(12);
        MI_TRY(build_table(ctx, &st->small_f, 2048, w4096, Fr::one(), 0));
        MI_TRY(build_table(ctx, &st->small_i, 2048, fe_inv(w4096), Fr::one(), 0));
        Fr w64k = domain_generator(16);
        MI_TRY(build_table(ctx, &st->tw64k_f, 65536, w64k, Fr::one(), 0));
        MI_TRY(build_table(ctx, &st->tw64k_i, 65536, fe_inv(w64k), Fr::one(), 0));
    }
    if (st->log_n == log_n) return MI_OK;
    Fr w = domain_generator(log_n), wi = fe_inv(w);
    Fr g = fe_from_u32<FrParams>(5), gi = fe_inv(g);
    Fr nn = Fr::zero();
    nn.l[0] = (u32)(1u << log_n);  // log_n <= 28
    Fr ninv = fe_inv(fe_to_mont(nn));
    u32 h = (log_n + 1) / 2;
    u32 nlo = 1u << h, nhi = 1u << (log_n - h);
    st->tw_h = h;
    MI_TRY(build_table(ctx, &st->tw_lo_f, nlo, w, Fr::one(), 0));
    MI_TRY(build_table(ctx, &st->tw_hi_f, nhi, w, Fr::one(), h));
    MI_TRY(build_table(ctx, &st->tw_lo_i, nlo, wi, Fr::one(), 0));
    MI_TRY(build_table(ctx, &st->tw_hi_i, nhi, wi, Fr::one(), h));
    MI_TRY(build_table(ctx, &st->g_lo, nlo, g, Fr::one(), 0));
    MI_TRY(build_table(ctx, &st->g_hi, nhi, g, Fr::one(), h));
    MI_TRY(build_table(ctx, &st->gi_lo, nlo, gi, Fr::one(), 0));
    MI_TRY(build_table(ctx, &st->gi_hi, nhi, gi, ninv, h));
    MI_TRY(build_table(ctx, &st->ninv, 1, Fr::one(), ninv, 0));
    // computeH folds the 1/N of FFTInverse(a|b|c) into the coset shift of the next transform and den = 1/(g^N - 1) into the
    // last inverse transform's scaling (both are linear): 4 of its 121 products per element disappear
    Fr gn = g;
    for (u32 k = 0; k < log_n; k++) gn = fe_sqr(gn);
    Fr den = fe_inv(gn - Fr::one());
    MI_TRY(build_table(ctx, &st->g_hi_n, nhi, g, ninv, h));
    MI_TRY(build_table(ctx, &st->gi_hi_nd, nhi, gi, ninv * den, h));
    MI_TRY(build_table(ctx, &st->ninv_den, 1, Fr::one(), ninv * den, 0));
    st->log_n = log_n;
    return MI_OK;
}

// computeH's direct tables for (log_n, current plan); a failure to allocate them only means the composed-table path runs
static int32_t ensure_direct(mi_ctx *ctx, u32 log_n) {
    NttState *st = state_of(ctx);
    const NttKnobs kn = knobs_for(st, log_n);
    const NttPlan pl = ntt_make_plan(log_n, kn.max_contig, kn.max_strided);
    if (st->d_log_n == log_n && st->d_log_r0 == pl.log_r[0] && st->d_npass == pl.n_pass) return MI_OK;
    Fr **four[] = {&st->d_tw_inv, &st->d_tw_fwd, &st->d_sc_fwd, &st->d_sc_inv};
    for (Fr **p : four) if (*p) { (void)hipFree(*p); *p = nullptr; }
    st->d_log_n = 0xffffffffu;
    if (log_n < st->direct_min_log_n) return MI_OK;
    MI_TRY(ensure_tables(ctx, log_n));
    const size_t bytes = sizeof(Fr) << log_n;
    for (Fr **p : four)
        if (hipMalloc((void **)p, bytes) != hipSuccess) {   // no room: stay on the composed tables
            (void)hipGetLastError();
            for (Fr **q : four) if (*q) { (void)hipFree(*q); *q = nullptr; }
            return MI_OK;
        }
    const Fr w = domain_generator(log_n), wi = fe_inv(w), g = fe_from_u32<FrParams>(5), gi = fe_inv(g);
    Fr nn = Fr::zero();
    nn.l[0] = (u32)(1u << log_n);
    const Fr ninv = fe_inv(fe_to_mont(nn));
    Fr gn = g;
    for (u32 k = 0; k < log_n; k++) gn = fe_sqr(gn);
    const Fr den = fe_inv(gn - Fr::one());
    const unsigned blocks = (unsigned)(((size_t)1 << log_n) + 255) / 256;
    if (pl.n_pass > 1) {
        hipLaunchKernelGGL(k_tw_layout, dim3(blocks), dim3(256), 0, ctx->stream, st->d_tw_inv, log_n, pl.log_r[0], wi);
        hipLaunchKernelGGL(k_tw_layout, dim3(blocks), dim3(256), 0, ctx->stream, st->d_tw_fwd, log_n, pl.log_r[0], w);
    }
    hipLaunchKernelGGL(k_sc_layout, dim3(blocks), dim3(256), 0, ctx->stream, st->d_sc_fwd, log_n, g, ninv);          // variant 2
    hipLaunchKernelGGL(k_sc_layout, dim3(blocks), dim3(256), 0, ctx->stream, st->d_sc_inv, log_n, gi, ninv * den);   // variant 3
    MI_CHECK_HIP(ctx, hipGetLastError());
    st->d_log_n = log_n; st->d_log_r0 = pl.log_r[0]; st->d_npass = pl.n_pass;
    return MI_OK;
}

// One transform of size 2^log_n: dst <- NTT(src[0..n_valid) zero padded).  dst == src allowed.
// variant (computeH only): 1 = inverse without the 1/N scaling; 2 = forward coset with 1/N folded into the shift tables;
// 3 = inverse coset with den folded into its scaling; 4 = inverse scaled by den / N.  0 = exactly fft.Domain's FFT / FFTInverse.
// load_mul / store_sub (computeH only, see NttPass): pointwise factor on the way into the first pass / pointwise subtrahend on the
// way out of the last.
// skip (computeH's fused contiguous pair, k_ntt_contig_pair): bit 0 = the first pass, bit 1 = the last pass is not launched but
// returned in *cap / *cap_t for the fused kernel; bit 2 = nothing is launched at all (capture only).
static int32_t ntt_run(mi_ctx *ctx, Fr *dst, const Fr *src, u32 n_valid, u32 log_n, u32 flags, u32 variant = 0,
                       const Fr *load_mul = nullptr, const Fr *store_sub = nullptr, u32 skip = 0, NttPass *cap = nullptr, NttTables *cap_t = nullptr) {
    NttState *st = state_of(ctx);
    MI_TRY(ensure_tables(ctx, log_n));
    const bool inverse = flags & MI_NTT_INVERSE, coset = flags & MI_NTT_COSET, dit = flags & MI_NTT_DIT;
    NttTables t{};
    t.small = inverse ? st->small_i : st->small_f;
    t.tw_lo = inverse ? st->tw_lo_i : st->tw_lo_f;
    t.tw_hi = inverse ? st->tw_hi_i : st->tw_hi_f;
    t.tw_h = st->tw_h;
    t.tw_64k = inverse ? st->tw64k_i : st->tw64k_f;
    u32 load_scale = 0, store_scale = 0;
    if (coset && !inverse) { t.sc_lo = st->g_lo; t.sc_hi = variant == 2 ? st->g_hi_n : st->g_hi; load_scale = dit ? 1 : 2; }
    else if (coset && inverse) { t.sc_lo = st->gi_lo; t.sc_hi = variant == 3 ? st->gi_hi_nd : st->gi_hi; store_scale = dit ? 4 : 3; }
    else if (inverse && variant != 1) { t.sc_lo = t.sc_hi = variant == 4 ? st->ninv_den : st->ninv; store_scale = 5; }
    if (load_mul && (load_scale || dit)) MI_FAIL(ctx, MI_EINVAL, "internal: load_mul on a pass whose load side already has a factor");

    const NttKnobs kn = knobs_for(st, log_n);
    NttPlan pl = ntt_make_plan(log_n, kn.max_contig, kn.max_strided);
    // log_s of pass i (DIF order): product of later radices
    u32 log_s[8];
    for (u32 i = 0, acc = log_n; i < pl.n_pass; i++) { acc -= pl.log_r[i]; log_s[i] = acc; }
    const Fr *cur_src = src;
    for (u32 step = 0; step < pl.n_pass; step++) {
        u32 i = dit ? pl.n_pass - 1 - step : step;
        NttPass p{};
        p.log_n = log_n; p.log_r = pl.log_r[i]; p.log_s = log_s[i];
        u32 room = kn.log_e > p.log_r ? kn.log_e - p.log_r : 0;
        u32 avail = p.log_s == 0 ? log_n - p.log_r : p.log_s;
        p.log_c = room < avail ? room : avail;
        p.dit = dit ? 1 : 0;
        p.twiddle = p.log_s != 0;
        p.scale = 0;
        if (step == 0 && load_scale) p.scale = load_scale;
        if (step == pl.n_pass - 1 && store_scale) {
            if (p.scale) MI_FAIL(ctx, MI_EINVAL, "internal: load and store scale on one pass");
            p.scale = store_scale;
        }
        p.n_valid = step == 0 ? n_valid : (1u << log_n);
        if (step == 0) p.load_mul = load_mul;
        if (step == pl.n_pass - 1) p.store_sub = store_sub;
        // the elements travel lazily reduced (below 4p) between the stages and the passes (ntt_tile.cuh); the last pass of a transform
        // whose output is final -- fft.Domain's own transforms (variant 0) and computeH's last one (variant 3: h) -- stores canonical values
        if (step == pl.n_pass - 1 && (variant == 0 || variant == 3)) p.canon = 1;
        // computeH's direct tables (variant != 0 only): the twiddle of the M = N pass, the coset shift of the contiguous pass
        const bool direct = variant != 0 && st->d_log_n == log_n && st->d_sc_fwd;
        if (direct && p.twiddle && i == 0 && pl.n_pass > 1) p.tw_direct = inverse ? st->d_tw_inv : st->d_tw_fwd;
        if (direct && variant == 2 && p.scale == 1) p.sc_direct = st->d_sc_fwd;
        if (direct && variant == 3 && p.scale == 3) p.sc_direct = st->d_sc_inv;
        const bool wave = p.log_r >= 7 && st->wave_stages;
        p.lds_pad = wave && p.log_s != 0 ? 1 : 0;
        if (((skip & 1) && step == 0) || ((skip & 2) && step == pl.n_pass - 1)) {
            if (cap) { *cap = p; *cap_t = t; }
            cur_src = dst;
            continue;
        }
        if (skip & 4) continue;
        u32 tiles = 1u << (log_n - p.log_r - p.log_c);
        size_t lds_bytes = (size_t)32 * ntt_plane_slots(p);
        if (st->lds_floor > lds_bytes) lds_bytes = st->lds_floor;   // occupancy cap of the pass kernels (mi_debug_set_knob "ntt_lds_floor_kb")
        u32 E = 1u << (p.log_r + p.log_c);
        // one butterfly per thread per stage when the tile allows: 64 KiB of LDS admits two workgroups per CU,
        // so 1024-thread workgroups are what fills the SIMDs (8 waves each) and hides the mad->addc chains
        u32 threads = E / 2 >= st->threads ? st->threads : (E / 2 >= 64 ? E / 2 : 64);
        if (wave) hipLaunchKernelGGL(k_ntt_pass_wave, dim3(tiles), dim3(threads > 256 ? 256 : threads), lds_bytes, ctx->stream, dst, cur_src, p, t);
        else hipLaunchKernelGGL(k_ntt_pass, dim3(tiles), dim3(threads), lds_bytes, ctx->stream, dst, cur_src, p, t);
        MI_CHECK_HIP(ctx, hipGetLastError());
        cur_src = dst;
        ctx->stats.ntt_launches++;
    }
    if (!(skip & 4)) ctx->stats.ntt_elems += (u64)1 << log_n;
    return MI_OK;
}

int32_t mi_ntt_dev_impl(mi_ctx *ctx, mi_fr *inout_dev, uint32_t log_n, uint32_t flags) {
    return ntt_run(ctx, (Fr *)inout_dev, (const Fr *)inout_dev, 1u << log_n, log_n, flags);
}

// computeH in four parts on ctx->stream, so that a caller whose inputs arrive one vector at a time (the host-pointer prove: a, b, c cross
// PCIe one after the other) can start a's transforms while b is still on the bus:
//   part 0 (src = a), part 1 (src = b)   N FFTInverse(., DIF) and the coset FFT up to (not including) its last pass when that pass runs in
//                                        the strided triple, else all of it; a lives in h_out, b in ws[0]
//   part 2 (src = c)                     den FFTInverse(c, DIF) into ws[1]; src2 != null: c is not given and is formed as src o src2 (the ORIGINAL a and b,
//                                        row by row) on the way into the first pass -- what c is for every witness gnark's solver accepts
//   part 3                               the strided triple (or the coset FFTs' last passes) and the last transform -> h_out
// The same launches as the one-call form, in an order that differs only between independent vectors: identical h.
struct ComputeHPlan {
    bool pair, triple, last;   // last: c's last pass and the last transform's last pass run fused (decided from the plan alone: both are the
                               // contiguous pass of a plain DIF transform, equal shapes by construction)
    NttPass pc, plast;
    NttTables tc, tlast;
};
static int32_t compute_h_plan(mi_ctx *ctx, uint32_t log_n, Fr *A, ComputeHPlan &cp) {
    const size_t n = (size_t)1 << log_n;
    NttState *st = state_of(ctx);
    const NttKnobs kn = knobs_for(st, log_n);
    const NttPlan pl = ntt_make_plan(log_n, kn.max_contig, kn.max_strided);
    // a and b: the contiguous last pass of the inverse transform and the contiguous first pass of the coset transform run as ONE
    // launch on the same tiles (k_ntt_contig_pair) when the plan has such a pair of radix >= 2^7
    cp.pair = st->fuse_pair && st->wave_stages && pl.n_pass >= 2 && pl.log_r[pl.n_pass - 1] >= 7;
    // The coset FFT's LAST pass (of a and of b), the product and the last transform's FIRST pass share their strided tiles: one
    // launch (k_ntt_strided_triple / k_ntt_strided_triple8) when that pass has radix 2^7 or 2^8 and the tile shape fits
    cp.triple = st->fuse_triple && st->wave_stages && pl.n_pass >= 2 && (pl.log_r[0] == 7 || pl.log_r[0] == 8);
    {
        const u32 lr = pl.log_r[pl.n_pass - 1];
        const u32 room = kn.log_e > lr ? kn.log_e - lr : 0, avail = log_n - lr, lc = room < avail ? room : avail;
        cp.last = st->fuse_last && st->wave_stages && pl.n_pass >= 2 && lr >= 7 && (1u << (lr + lc)) <= 4 * 256;
    }
    cp.pc = NttPass{}; cp.plast = NttPass{}; cp.tc = NttTables{}; cp.tlast = NttTables{};
    if (cp.triple) {   // capture the two passes without launching anything, and check the tile shape before committing to the fused form
        MI_TRY(ntt_run(ctx, A, A, (u32)n, log_n, MI_NTT_DIT | MI_NTT_COSET, 2, nullptr, nullptr, 2 | 4, &cp.pc, &cp.tc));
        MI_TRY(ntt_run(ctx, A, A, (u32)n, log_n, MI_NTT_INVERSE | MI_NTT_COSET, 3, nullptr, nullptr, 1 | 4, &cp.plast, &cp.tlast));
        const NttPass &pc = cp.pc, &plast = cp.plast;
        cp.triple = pc.log_r == plast.log_r && pc.log_s == plast.log_s && pc.log_c == plast.log_c && pc.log_s != 0 && pc.lds_pad == plast.lds_pad && !pc.scale && !plast.scale &&
                    ((pc.log_r == 7 && pc.log_c <= 3) ||                      // one wave per 128-row sub-block, up to 512 threads
                     (pc.log_r == 8 && pc.log_c >= 1 && pc.log_c <= 2));      // k_ntt_strided_triple8: 256 threads, 1 or 2 pairs at distance 128 each
    }
    return MI_OK;
}
int32_t mi_compute_h_part(mi_ctx *ctx, uint32_t log_n, int part, const mi_fr *src, size_t n_constraints, mi_fr *h_out, const mi_fr *src2) {
    static const char *const range_names[4] = {"mi.computeH.a.enqueue", "mi.computeH.b.enqueue", "mi.computeH.c.enqueue", "mi.computeH.last.enqueue"};
    const MiRange range(range_names[part & 3]);
    const size_t n = (size_t)1 << log_n;
    // gnark's computeH (7 transforms): a, b, c <- FFTInverse; a, b, c <- FFT on the coset; a <- (a b - c) den; h <- FFTInverse on
    // the coset.  The last transform is linear and undoes the coset FFT of c exactly:
    //     h = cosetFFTInverse((ca cb - cc) den) = den cosetFFTInverse(ca cb) - den FFTInverse(c),
    // so the coset FFT of c is never computed -- SIX transforms, the same field elements (exact arithmetic), for ANY a, b, c.
    MI_TRY(ensure_direct(ctx, log_n));
    MI_TRY(mi_reserve(ctx, ctx->ws[0], n * sizeof(Fr)));
    MI_TRY(mi_reserve(ctx, ctx->ws[1], n * sizeof(Fr)));
    Fr *A = (Fr *)h_out, *B = (Fr *)ctx->ws[0].p, *C = (Fr *)ctx->ws[1].p;
    ComputeHPlan cp;
    MI_TRY(compute_h_plan(ctx, log_n, A, cp));
    const u32 first_skip = cp.pair ? 1u : 0u;
    if (part == 0 || part == 1) {
        // 1. v <- N FFTInverse(v, DIF) (zero padding fused into the first pass's load; the 1/N rides in the coset shift)
        Fr *v = part == 0 ? A : B;
        if (!cp.pair) {
            MI_TRY(ntt_run(ctx, v, (const Fr *)src, (u32)n_constraints, log_n, MI_NTT_INVERSE, 1));
        } else {
            NttPass pi{}, pf{};
            NttTables ti{}, tf{};
            MI_TRY(ntt_run(ctx, v, (const Fr *)src, (u32)n_constraints, log_n, MI_NTT_INVERSE, 1, nullptr, nullptr, 2, &pi, &ti));
            MI_TRY(ntt_run(ctx, v, v, (u32)n, log_n, MI_NTT_DIT | MI_NTT_COSET, 2, nullptr, nullptr, 1 | 4, &pf, &tf));
            if (pi.log_r != pf.log_r || pi.log_c != pf.log_c || pi.log_s || pf.log_s) MI_FAIL(ctx, MI_EINVAL, "internal: contiguous pass pair does not match");
            hipLaunchKernelGGL(k_ntt_contig_pair, dim3(1u << (log_n - pi.log_r - pi.log_c)), dim3(256), (size_t)32 * ntt_plane_slots(pi), ctx->stream, v, pi, ti, pf, tf);
            MI_CHECK_HIP(ctx, hipGetLastError());
            ctx->stats.ntt_launches++;
        }
        // 2. v <- FFT(v, DIT, OnCoset): the rest of it after the pair, without its last pass when that one runs in the triple
        NttPass px{};
        NttTables tx{};
        return ntt_run(ctx, v, v, (u32)n, log_n, MI_NTT_DIT | MI_NTT_COSET, 2, nullptr, nullptr, first_skip | (cp.triple ? 2u : 0u), &px, &tx);
    }
    if (part == 2) {   // c <- den FFTInverse(c, DIF) (one constant den / N on the way out), bit-reversed like h; its last pass waits for part 3 when fused
        NttPass px{};
        NttTables tx{};
        return ntt_run(ctx, C, (const Fr *)src, (u32)n_constraints, log_n, MI_NTT_INVERSE, 4, (const Fr *)src2, nullptr, cp.last ? 2u : 0u, &px, &tx);
    }
    // 3. h <- den FFTInverse(a b, DIF, OnCoset) - c, left bit-reversed like gnark: the product a b is taken on the way into the
    //    first pass, the subtraction on the way out of the last (no pointwise kernel, no extra round trip through HBM)
    // the last launch when fused: both last passes captured (nothing launched), one kernel does them and the subtraction
    auto last_fused = [&]() -> int32_t {
        NttPass pa{}, pcl{};
        NttTables ta{}, tcl{};
        MI_TRY(ntt_run(ctx, C, C, (u32)n, log_n, MI_NTT_INVERSE, 4, nullptr, nullptr, 2 | 4, &pcl, &tcl));
        MI_TRY(ntt_run(ctx, A, A, (u32)n, log_n, MI_NTT_INVERSE | MI_NTT_COSET, 3, nullptr, nullptr, 2 | 4, &pa, &ta));
        if (pa.log_r != pcl.log_r || pa.log_c != pcl.log_c || pa.log_s || pcl.log_s || pa.dit || pcl.dit || pa.lds_pad != pcl.lds_pad)
            MI_FAIL(ctx, MI_EINVAL, "internal: the two last passes of computeH do not match");
        // the Z MSM's digit count rides along when prove.hip armed the hook and this launch's tiles are whole slices of that sort
        const u32 E = 1u << (pa.log_r + pa.log_c);
        Msm2Shape zs{};
        bool count = ctx->zhook.armed && ctx->zhook.C1;
        if (count) {
            std::memcpy(&zs, ctx->zhook.shape, sizeof(zs));
            count = E % MSM2_SLICE == 0 && E / MSM2_SLICE <= 4 && zs.n <= n && (u64)zs.nslices * MSM2_SLICE == n && zs.c >= 17 && zs.c <= 22 && !zs.wkeys &&
                    (size_t)32 * ntt_plane_slots(pa) + (size_t)(E / MSM2_SLICE) * zs.ngroups * 4 <= 160 * 1024;
        }
        ctx->zhook.armed = false;
        if (count) {
            const size_t lds_bytes = (size_t)32 * ntt_plane_slots(pa) + (size_t)(E / MSM2_SLICE) * zs.ngroups * 4;
#define MI_LAST_COUNT(CW) do { \
            hipLaunchKernelGGL(k_ntt_contig_last_sub_count<CW>, dim3(1u << (log_n - pa.log_r - pa.log_c)), dim3(256), lds_bytes, ctx->stream, A, (const Fr *)C, pa, ta, pcl, tcl, zs, ctx->zhook.C1); } while (0)
            switch (zs.c) { case 17: MI_LAST_COUNT(17); break; case 18: MI_LAST_COUNT(18); break; case 19: MI_LAST_COUNT(19); break; case 20: MI_LAST_COUNT(20); break;
                            case 21: MI_LAST_COUNT(21); break; default: MI_LAST_COUNT(22); break; }
#undef MI_LAST_COUNT
            ctx->zhook.done = true; ctx->zhook.h = A;
            ctx->z_count_fused_launches++;
        } else {
            hipLaunchKernelGGL(k_ntt_contig_last_sub, dim3(1u << (log_n - pa.log_r - pa.log_c)), dim3(256), (size_t)32 * ntt_plane_slots(pa), ctx->stream, A, (const Fr *)C, pa, ta, pcl, tcl);
        }
        MI_CHECK_HIP(ctx, hipGetLastError());
        ctx->stats.ntt_launches++;
        return MI_OK;
    };
    if (!cp.triple) {
        if (!cp.last) return ntt_run(ctx, A, A, (u32)n, log_n, MI_NTT_INVERSE | MI_NTT_COSET, 3, B, C);
        NttPass px{};
        NttTables tx{};
        MI_TRY(ntt_run(ctx, A, A, (u32)n, log_n, MI_NTT_INVERSE | MI_NTT_COSET, 3, B, nullptr, 2, &px, &tx));
        return last_fused();
    }
    // radix 2^7: one wave per 128-row sub-block of the tile, 64 * 2^log_c threads (256 at the default 2^9-element tiles); radix 2^8: 256 threads
    const NttPass &pc = cp.pc;
    if (pc.log_r == 7)
        hipLaunchKernelGGL(k_ntt_strided_triple, dim3(1u << (log_n - pc.log_r - pc.log_c)), dim3(64u << pc.log_c), (size_t)32 * ntt_plane_slots(pc), ctx->stream,
                           A, (const Fr *)B, pc, cp.tc, cp.plast, cp.tlast);
    else
        hipLaunchKernelGGL(k_ntt_strided_triple8, dim3(1u << (log_n - pc.log_r - pc.log_c)), dim3(256), (size_t)32 * ntt_plane_slots(pc), ctx->stream,
                           A, (const Fr *)B, pc, cp.tc, cp.plast, cp.tlast);
    MI_CHECK_HIP(ctx, hipGetLastError());
    ctx->stats.ntt_launches++;
    if (!cp.last) return ntt_run(ctx, A, A, (u32)n, log_n, MI_NTT_INVERSE | MI_NTT_COSET, 3, nullptr, C, 1);
    NttPass px{};
    NttTables tx{};
    MI_TRY(ntt_run(ctx, A, A, (u32)n, log_n, MI_NTT_INVERSE | MI_NTT_COSET, 3, nullptr, nullptr, 1 | 2, &px, &tx));
    return last_fused();
}
int32_t mi_compute_h_dev_impl(mi_ctx *ctx, uint32_t log_n, const mi_fr *a, const mi_fr *b, const mi_fr *c,
                              size_t n_constraints, mi_fr *h_out) {
    MI_TRY(mi_compute_h_part(ctx, log_n, 0, a, n_constraints, h_out));
    MI_TRY(mi_compute_h_part(ctx, log_n, 1, b, n_constraints, h_out));
    if (c) MI_TRY(mi_compute_h_part(ctx, log_n, 2, c, n_constraints, h_out));
    else MI_TRY(mi_compute_h_part(ctx, log_n, 2, a, n_constraints, h_out, b));   // c = a o b, formed on the device
    return mi_compute_h_part(ctx, log_n, 3, nullptr, n_constraints, h_out);
}

static void stats_begin(mi_ctx *ctx) { std::memset(&ctx->stats, 0, sizeof(ctx->stats)); }

extern "C" {
int32_t mi_debug_set_ntt_plan(mi_ctx *ctx, uint32_t log_e, uint32_t max_contig, uint32_t max_strided) {
    if (!ctx || log_e < 1 || log_e > 12 || max_contig < 1 || max_contig > 12 || max_contig > log_e || max_strided < 1 || max_strided > 12 || max_strided > log_e)
        return MI_EINVAL;
    NttState *st = state_of(ctx);
    st->log_e = log_e; st->max_contig = max_contig; st->max_strided = max_strided; st->plan_set = 1;
    return MI_OK;
}
int32_t mi_debug_set_ntt_wave_stages(mi_ctx *ctx, uint32_t on, uint32_t direct_min_log_n) {
    if (!ctx || on > 1) return MI_EINVAL;
    NttState *st = state_of(ctx);
    st->wave_stages = on; st->direct_min_log_n = direct_min_log_n;
    st->d_npass = 0;   // force the next computeH to re-decide about its direct tables
    return MI_OK;
}
int32_t mi_debug_set_ntt_fuse_pair(mi_ctx *ctx, uint32_t on) {
    if (!ctx || on > 7) return MI_EINVAL;
    state_of(ctx)->fuse_pair = on & 1u;          // bit 0: the contiguous pair (k_ntt_contig_pair)
    state_of(ctx)->fuse_triple = (on >> 1) & 1u;  // bit 1: the strided triple (k_ntt_strided_triple)
    state_of(ctx)->fuse_last = (on >> 2) & 1u;    // bit 2: c's last pass + the last transform's last pass (k_ntt_contig_last_sub)
    return MI_OK;
}
int32_t mi_debug_set_ntt_threads(mi_ctx *ctx, uint32_t threads) {
    if (!ctx || threads < 64 || threads > 1024 || (threads & (threads - 1))) return MI_EINVAL;
    state_of(ctx)->threads = threads;
    return MI_OK;
}
int32_t mi_ntt_dev(mi_ctx *ctx, mi_fr *inout_dev, uint32_t log_n, uint32_t flags) {
    if (!ctx || !inout_dev || log_n > 28 || (flags & ~7u)) return MI_EINVAL;
    stats_begin(ctx);
    MI_CHECK_HIP(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
    MI_TRY(mi_ntt_dev_impl(ctx, inout_dev, log_n, flags));
    MI_CHECK_HIP(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
    MI_CHECK_HIP(ctx, hipEventSynchronize(ctx->ev[1]));
    MI_CHECK_HIP(ctx, hipEventElapsedTime(&ctx->stats.ntt_kernel_ms, ctx->ev[0], ctx->ev[1]));
    return MI_OK;
}
int32_t mi_ntt(mi_ctx *ctx, mi_fr *inout, uint32_t log_n, uint32_t flags) {
    if (!ctx || !inout || log_n > 28 || (flags & ~7u)) return MI_EINVAL;
    size_t bytes = sizeof(Fr) << log_n;
    MI_TRY(mi_reserve(ctx, ctx->ws[2], bytes));
    MI_CHECK_HIP(ctx, hipMemcpyAsync(ctx->ws[2].p, inout, bytes, hipMemcpyHostToDevice, ctx->stream));
    MI_TRY(mi_ntt_dev(ctx, (mi_fr *)ctx->ws[2].p, log_n, flags));
    MI_CHECK_HIP(ctx, hipMemcpyAsync(inout, ctx->ws[2].p, bytes, hipMemcpyDeviceToHost, ctx->stream));
    MI_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MI_OK;
}
int32_t mi_compute_h_dev(mi_ctx *ctx, uint32_t log_n, const mi_fr *a, const mi_fr *b, const mi_fr *c,
                         size_t n_constraints, mi_fr *h_out) {
    if (!ctx || !a || !b || !h_out || log_n > 28 || n_constraints > ((size_t)1 << log_n)) return MI_EINVAL;   // c may be null: c = a o b
    stats_begin(ctx);
    MI_CHECK_HIP(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
    MI_TRY(mi_compute_h_dev_impl(ctx, log_n, a, b, c, n_constraints, h_out));
    MI_CHECK_HIP(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
    MI_CHECK_HIP(ctx, hipEventSynchronize(ctx->ev[1]));
    MI_CHECK_HIP(ctx, hipEventElapsedTime(&ctx->stats.compute_h_ms, ctx->ev[0], ctx->ev[1]));
    ctx->stats.ntt_kernel_ms = ctx->stats.compute_h_ms;
    return MI_OK;
}
int32_t mi_compute_h(mi_ctx *ctx, uint32_t log_n, const mi_fr *a, const mi_fr *b, const mi_fr *c,
                     size_t n_constraints, mi_fr *h_out) {
    if (!ctx || !a || !b || !h_out || log_n > 28 || n_constraints > ((size_t)1 << log_n)) return MI_EINVAL;   // c may be null: c = a o b
    size_t nb = n_constraints * sizeof(Fr), full = sizeof(Fr) << log_n;
    MI_TRY(mi_reserve(ctx, ctx->ws[2], full));
    MI_TRY(mi_reserve(ctx, ctx->ws[3], nb * 3 + 96));
    char *in = (char *)ctx->ws[3].p;
    MI_CHECK_HIP(ctx, hipMemcpyAsync(in, a, nb, hipMemcpyHostToDevice, ctx->stream));
    MI_CHECK_HIP(ctx, hipMemcpyAsync(in + nb, b, nb, hipMemcpyHostToDevice, ctx->stream));
    if (c) MI_CHECK_HIP(ctx, hipMemcpyAsync(in + 2 * nb, c, nb, hipMemcpyHostToDevice, ctx->stream));
    MI_TRY(mi_compute_h_dev(ctx, log_n, (mi_fr *)in, (mi_fr *)(in + nb), c ? (mi_fr *)(in + 2 * nb) : nullptr, n_constraints, (mi_fr *)ctx->ws[2].p));
    MI_CHECK_HIP(ctx, hipMemcpyAsync(h_out, ctx->ws[2].p, full, hipMemcpyDeviceToHost, ctx->stream));
    MI_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MI_OK;
}
}
