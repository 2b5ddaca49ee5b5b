// Pippenger MSM on gfx950 for BN254 G1 and G2: kernels around the per-thread bodies of
// msm_core.cuh plus the (sync-free) host orchestration.  See msm_core.cuh for the algorithm and the
// reference functions it replaces (gnark-crypto MultiExp, reached from /root/reference/mt.go:496).
//
// HBM layout per MSM call (all grow-only ctx workspaces, n pairs, nwin windows, K = nwin*2^(c-1) keys):
//   digits  int16 [nwin][n]            window-major signed digits
//   H       u32   [nwin][G][2^(c-1)]   per-(window, slice) bucket histogram, then its exclusive prefix along the slices
//   keystart u32  [K+1]                exclusive scan of the per-key totals
//   sorted  u32   [<= nwin*n]          point index | sign<<31, grouped by key
//   start/cnt/items/item_start u32 [K] item decomposition of the current level (ping-pong)
//   partial XYZZ  [items]              per-item partial sums (ping-pong between levels)
//   bucket  XYZZ  [K]                  final bucket sums (zero-filled = infinity)
// Roofline: the level-1 accumulate kernel reads 4 B + one 64 B (G1) / 128 B (G2) point per entry and
// performs one mixed addition (~10 Fp products): it is VALU-bound by two orders of magnitude; the HBM
// figure reported for it is the algorithmic 96 B (160 B) per pair of SURVEY 8d over its duration.
#include "ctx.h"
#include "msm2_core.cuh"
#include "msm_curve_ops.h"
#include <cstring>
#include <cstdlib>
#include <new>

struct MsmKnobs {
    u32 c = 0, L1 = 0, L2 = 0, seg = 0, G = 0;  // 0 = automatic
    u32 no_rprime = 0;                          // 1: G1 level 1 stays on the 8 x 32-bit kernel everywhere (tests compare both)
    u32 ba_rounds = 0;                          // G1 level 1 by batch-affine rounds (msm_ba_g1.cuh): 0 = off, 1..4 rounds (MI_MSM_BA_ROUNDS)
    u32 l1_waves = 3;                           // G1 level-1 29-bit kernel: the build for 3 (default) or 2 waves per SIMD (msm_g1.hip)
    u32 precompute_unbatched = 0;               // 1: window tables by the one-kernel form (one inversion per point per window)
    u32 std_partials = 0;                       // 1: partial sums between the levels in the standard form even after a 29-bit level 1
    u32 one_pass_sort = 0;                      // 1: large generic MSMs keep the one-pass counting sort (tests compare both)
    u32 chunk = 0;                              // fixed-base sort: entries per pass-2 chunk (tests shrink it)
    u32 gbits = 0;                              // fixed-base sort: log2 buckets per pass-1 group (0 = automatic)
    u32 bound_levels = 0;                       // 1: as many item levels as the worst case needs (windows * n entries in one bucket) instead of
                                                // as many as the fullest bucket of THIS sort needs (tests compare both)
    // named knobs of mi_debug_set_knob (the header lists them): measurement switches that used to be environment variables
    u32 l1_wg = 4;                              // G1 level-1 29-bit kernel: waves per workgroup, 1 / 2 / 4 (4: +0.7..1.2 % proofs/s in 11 of 12 same-box pairs)
    u32 g2_wg = 1;                              // G2 level-1 29-bit kernel: waves per workgroup, 1 / 2 / 4
    u32 z_waves = 0;                            // 2: the Z MSM's level-1 launch alone on the two-waves-per-SIMD build
    u32 g1_grid_per_cu = 0, g2_grid_per_cu = 0; // resident-grid cap per CU of the level-1 launches (0 = 128)
    u32 count_per = 0;                          // fixed-base sort: slices per counting workgroup (0 = 32)
    u32 plain_scatter = 0;                      // fixed-base sort: 1 = pass 2 by the plain scatter instead of the staged one
    u32 dense_L1 = 0;                           // level-1 item size of a DENSE sort (>= half of the scalars' digits non-zero: a witness of mostly full-width values): 0 = automatic (32), 1 = off (the plan's L1), 4..64 = forced
    u32 flat_L1 = 0;                            // level-1 item size of a FLAT sort (fullest bucket <= 2 x the average: uniform scalars, prove's Z MSM): 0 = automatic
                                                // (msm_accum_enqueue: the average / L2^k that falls into 17..32, so that the levels above are full L2-ary trees), 1 = off (L1), 4..64 = forced
    u32 z_count_fused = 1;                      // 1: prove's Z MSM takes its digit count from computeH's last launch (ctx->zhook) instead of a count pass of its own
                                                // (built for VERDICT r4; throughput equal -- 33.96 against 33.97 proofs/s over 8 same-process rounds --, one proof alone 0.2 ms shorter)
    u32 finisher = 1;                           // 1: item levels whose fullest key holds <= finisher_max partial sums end in ONE launch (k_msm_finish_keys)
    u32 finisher_max = 0;                       // 0 = automatic
    u32 finisher_min_level = 2;                 // the finisher may follow accumulate pass number finisher_min_level + 1 at the earliest
};
static_assert(sizeof(MsmKnobs) <= sizeof(mi_ctx::msm_knobs), "mi_ctx::msm_knobs is too small");
static MsmKnobs *knobs_of(mi_ctx *ctx) { return reinterpret_cast<MsmKnobs *>(ctx->msm_knobs); }
__global__ void k_msm_hist(MsmShape s, const int16_t *digits, u32 *H);
__global__ void k_msm_scatter(MsmShape s, const int16_t *digits, const u32 *keystart, const u32 *Hx, u32 *sorted);
struct Msm2Shape;
__global__ void k_msm2_hist2(Msm2Shape s, const u32 *gstart, const u32 *cstart, const uint16_t *part_lo, u32 *H2);
__global__ void k_msm2_scatter2(Msm2Shape s, const u32 *gstart, const u32 *cstart, const u32 *keystart, const u32 *H2x,
                                const uint16_t *part_lo, const u32 *part_val, u32 *sorted);

// ---------------------------------------------------------------- kernels
__global__ void k_msm_digits(MsmShape s, const Fr *scalars, int montgomery, int16_t *digits) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < s.n) msm_digits_body(s, scalars, montgomery != 0, digits, i);
}
__global__ void __launch_bounds__(1024) k_msm_hist(MsmShape s, const int16_t *digits, u32 *H) {
    extern __shared__ u32 lds_u32[];
    const u32 g = blockIdx.x, w = blockIdx.y;
    msm_hist_zero(s, lds_u32, threadIdx.x, blockDim.x);
    __syncthreads();
    msm_hist_count(s, digits, g, w, lds_u32, threadIdx.x, blockDim.x);
    __syncthreads();
    msm_hist_write(s, H, g, w, lds_u32, threadIdx.x, blockDim.x);
}
__global__ void k_msm_colsum(MsmShape s, u32 *H, u32 *total) {
    u32 key = blockIdx.x * blockDim.x + threadIdx.x;
    if (key < s.nkeys) msm_colsum_body(s, H, total, key);
}
__global__ void __launch_bounds__(1024) k_msm_scatter(MsmShape s, const int16_t *digits, const u32 *keystart, const u32 *Hx, u32 *sorted) {
    extern __shared__ u32 lds_u32[];
    const u32 g = blockIdx.x, w = blockIdx.y;
    msm_scatter_init(s, keystart, Hx, g, w, lds_u32, threadIdx.x, blockDim.x);
    __syncthreads();
    msm_scatter_move(s, digits, g, w, lds_u32, sorted, threadIdx.x, blockDim.x);
}
// Also does what a hipMemsetAsync of the whole bucket array did (a launch of its own at the head of every accumulate stage, 67 MB for the Z
// MSM's 2^19 buckets): a key WITHOUT entries gets its bucket = infinity (all zero) here -- every other bucket is written by the item that
// ends it -- and the words behind the array (the finisher's list counters) are zeroed.  bucket_words = u32 words per bucket (32 / 64).
__global__ void k_msm_prep1(MsmShape s, const u32 *S, u32 L, u32 *start, u32 *cnt, u32 *items, u32 *bucket, u32 bucket_words) {
    u32 key = blockIdx.x * blockDim.x + threadIdx.x;
    if (key < s.nkeys) {
        msm_prep_level1(s, S, L, start, cnt, items, key);
        if (bucket && cnt[key] == 0) {
            uint4 *b = reinterpret_cast<uint4 *>(bucket + (size_t)key * bucket_words);
            for (u32 k = 0; k < bucket_words / 4; k++) b[k] = make_uint4(0, 0, 0, 0);
        }
    }
    if (bucket && key < 16) bucket[(size_t)s.nkeys * bucket_words + key] = 0;
}
__global__ void k_msm_prep_next(u32 nkeys, const u32 *prev_items, const u32 *prev_item_start, u32 L, u32 *start, u32 *cnt, u32 *items) {
    u32 key = blockIdx.x * blockDim.x + threadIdx.x;
    if (key < nkeys) msm_prep_next(prev_items, prev_item_start, L, start, cnt, items, key);
}
// ---------------------------------------------------------------- fixed-base MSM: two-pass bucket sort (msm2_core.cuh)
static constexpr u32 MSM2_MAX_GROUPS = 4096;   // 2^(c-1-gbits) <= 2^(21-9)
// workgroup = `per` consecutive slices (per * ngroups <= 8192 counters of LDS): the counters of a group are then written as
// per * 4-B contiguous segments of C1[group][slice] instead of single words 4 * nslices bytes apart
// (C: the window width at compile time -- the digits of a scalar then cost a few shifts of registers each, msm2_core.cuh; 0 = any width)
template <u32 C>
__global__ void __launch_bounds__(512) k_msm2_count(Msm2Shape s, const Fr *scalars, int montgomery, u32 per, u32 *C1) {
    __shared__ u32 lds[8192];
    const u32 s0 = blockIdx.x * per;
    const u32 cnt = s0 + per <= s.nslices ? per : s.nslices - s0;
    for (u32 k = threadIdx.x; k < per * s.ngroups; k += blockDim.x) lds[k] = 0;
    __syncthreads();
    msm2_count_slices<C>(s, scalars, montgomery != 0, s0, cnt, lds, threadIdx.x, blockDim.x);
    __syncthreads();
    for (u32 k = threadIdx.x; k < per * s.ngroups; k += blockDim.x) {
        u32 g = k / per, j = k % per;
        if (j < cnt) C1[(size_t)g * s.nslices + s0 + j] = lds[j * s.ngroups + g];
    }
}
// exclusive scan over the values of threads 0..255 of a larger workgroup (Hillis-Steele); every thread of the workgroup calls it
__device__ u32 block_exclusive_scan_first256(u32 v, u32 *lds, u32 *total) {
    const u32 tid = threadIdx.x;
    const bool in = tid < 256;
    if (in) lds[tid] = v;
    __syncthreads();
    for (u32 off = 1; off < 256; off <<= 1) {
        u32 a = in && tid >= off ? lds[tid - off] : 0;
        __syncthreads();
        if (in) lds[tid] += a;
        __syncthreads();
    }
    u32 incl = in ? lds[tid] : 0;
    *total = lds[255];
    __syncthreads();
    return incl - v;
}
static constexpr u32 MSM2_PART_THREADS = MSM2_SLICE;   // one scalar per thread: 8 waves per workgroup hide the LDS-atomic latency of place/copy
template <u32 C>
__global__ void __launch_bounds__(MSM2_PART_THREADS) k_msm2_partition(Msm2Shape s, const Fr *scalars, int montgomery, const u32 *S1, u32 cap,
                                                                      uint16_t *part_lo, u32 *part_val) {
    extern __shared__ u32 lds_u32[];
    u32 *hist = lds_u32, *loff = hist + s.ngroups, *gbase = loff + s.ngroups + 1, *tmp = gbase + s.ngroups, *stage_val = tmp + 256;
    uint16_t *stage_lo = (uint16_t *)(stage_val + cap), *stage_grp = stage_lo + cap;
    for (u32 g = threadIdx.x; g < s.ngroups; g += blockDim.x) hist[g] = 0;
    // a slice has at most MSM2_SLICE = blockDim scalars: one per thread, loaded and made canonical once, kept in registers for
    // both the count and the place phase
    // XCD-aware slice order (as in k_msm2_scatter2): XCD x walks the contiguous slices [x*Q, (x+1)*Q), so the runs that neighbouring
    // slices append to a group meet in ONE L2 and leave as whole lines
    const u32 Q = (s.nslices + 7) / 8, slice = (blockIdx.x & 7) * Q + (blockIdx.x >> 3);
    if ((blockIdx.x >> 3) >= Q || slice >= s.nslices) return;
    u32 begin, end;
    msm2_slice_range(s, slice, begin, end);
    const u32 i = begin + threadIdx.x;
    const bool have = i < end;
    Msm2Digits dg;
    if (have) dg.start(scalars[i], montgomery != 0);
    __syncthreads();
    if (have) msm2_count_one<C>(s, dg, hist);
    __syncthreads();
    u32 carry = 0;
    for (u32 base = 0; base < s.ngroups; base += 256) {   // loff = exclusive scan of hist; hist becomes the cursor
        u32 i = base + threadIdx.x;
        const bool mine = threadIdx.x < 256 && i < s.ngroups;
        u32 v = mine ? hist[i] : 0, total;
        u32 ex = block_exclusive_scan_first256(v, tmp, &total);
        if (mine) { loff[i] = carry + ex; hist[i] = carry + ex; gbase[i] = S1[(size_t)i * s.nslices + slice]; }
        carry += total;
    }
    if (threadIdx.x == 0) loff[s.ngroups] = carry;
    __syncthreads();
    if (have) msm2_place_one<C>(s, dg, i, hist, stage_lo, stage_val, stage_grp);
    __syncthreads();
    msm2_stage_copy_body(s, gbase, loff, stage_lo, stage_val, stage_grp, part_lo, part_val, threadIdx.x, blockDim.x);
}
__global__ void __launch_bounds__(1024) k_msm2_hist2(Msm2Shape s, const u32 *gstart, const u32 *cstart, const uint16_t *part_lo, u32 *H2) {
    extern __shared__ u32 lds_u32[];
    u32 hi, b, e;
    if (!msm2_chunk_range(s, gstart, cstart, blockIdx.x, hi, b, e)) return;   // the grid is a host-side bound
    msm2_hist2_zero(s, lds_u32, threadIdx.x, blockDim.x);
    __syncthreads();
    msm2_hist2_count(part_lo, b, e, lds_u32, threadIdx.x, blockDim.x);
    __syncthreads();
    msm2_hist2_write(s, H2, blockIdx.x, lds_u32, threadIdx.x, blockDim.x);
}
// also leaves the fullest bucket's size in *max_out (zeroed by k_msm2_chunk_count_scan earlier on the stream): what k_max_u32 computes for the one-pass sort
// key_block_sums != null: the wave's sum of totals is added to the key scan's block sum it belongs to (SCAN_BLOCK consecutive keys per block;
// zeroed by k_msm2_chunk_count_scan) -- the first kernel of that scan (k_scan_block_sums) is then not launched
__global__ void __launch_bounds__(256) k_msm2_colsum(Msm2Shape s, const u32 *cstart, u32 *H2, u32 *total, u32 *max_out, u32 *key_block_sums) {
    u32 key = blockIdx.x * blockDim.x + threadIdx.x;
    const u32 t = key < s.nkeys ? msm2_colsum_body(s, cstart, H2, total, key) : 0u;
    u32 m = t, sum = t;
    for (int off = 32; off > 0; off >>= 1) { const u32 o = (u32)__shfl_xor((int)m, off); m = o > m ? o : m; sum += (u32)__shfl_xor((int)sum, off); }
    if ((threadIdx.x & 63) == 0 && m) atomicMax(max_out, m);
    if (key_block_sums && (threadIdx.x & 63) == 0 && sum) atomicAdd(&key_block_sums[key / (16 * 64)], sum);   // (a wave's 64 keys lie in one block of 1024)
}
__global__ void __launch_bounds__(1024) k_msm2_scatter2(Msm2Shape s, const u32 *gstart, const u32 *cstart, const u32 *keystart, const u32 *H2x,
                                                        const uint16_t *part_lo, const u32 *part_val, u32 *sorted) {
    extern __shared__ u32 lds_u32[];
    // XCD-aware chunk order: blocks with equal blockIdx % 8 share an XCD (observed round-robin placement; a speed choice only).
    // XCD x walks the contiguous chunk range [x*Q, (x+1)*Q): its resident blocks then scatter into the same two or three
    // groups' output windows (1.7 MB each at 2^19 buckets / 2^23 pairs), which stay in that XCD's 4 MB L2 until their lines
    // are complete -- instead of every 4-B store of the launch going out as its own partial line.
    const u32 total = cstart[s.ngroups], Q = (total + 7) / 8;
    const u32 chunk_id = (blockIdx.x & 7) * Q + (blockIdx.x >> 3);
    if ((blockIdx.x >> 3) >= Q) return;
    u32 hi, b, e;
    if (!msm2_chunk_range(s, gstart, cstart, chunk_id, hi, b, e)) return;
    msm2_scatter2_init(s, keystart, H2x, chunk_id, hi, lds_u32, threadIdx.x, blockDim.x);
    __syncthreads();
    msm2_scatter2_move(part_lo, part_val, b, e, lds_u32, sorted, threadIdx.x, blockDim.x);
}

// The staged scatter (msm2_core.cuh): LDS = cursor[gsize] | cnt -> loc [gsize] | wave sums [16] | stage values [chunk] | stage buckets [chunk] (u16).
// Needs chunk <= 8 * 1024 entries; the launcher falls back to k_msm2_scatter2 otherwise.
__global__ void __launch_bounds__(1024) k_msm2_scatter2_staged(Msm2Shape s, const u32 *gstart, const u32 *cstart, const u32 *keystart, const u32 *H2x,
                                                               const uint16_t *part_lo, const u32 *part_val, u32 *sorted) {
    extern __shared__ u32 lds_u32[];
    u32 *cursor = lds_u32, *loc = cursor + s.gsize, *wsum = loc + s.gsize, *st_val = wsum + 16;
    uint16_t *st_lo = (uint16_t *)(st_val + s.chunk);
    const u32 total = cstart[s.ngroups], Q = (total + 7) / 8;
    const u32 chunk_id = (blockIdx.x & 7) * Q + (blockIdx.x >> 3);   // XCD-aware chunk order, as in k_msm2_scatter2
    if ((blockIdx.x >> 3) >= Q) return;
    u32 hi, b, e;
    if (!msm2_chunk_range(s, gstart, cstart, chunk_id, hi, b, e)) return;
    msm2_scatter2_init(s, keystart, H2x, chunk_id, hi, cursor, threadIdx.x, blockDim.x);
    for (u32 k = threadIdx.x; k < s.gsize; k += blockDim.x) loc[k] = 0;
    __syncthreads();
    uint16_t lo[MSM2_STAGE_PER];
    u32 val[MSM2_STAGE_PER], rank[MSM2_STAGE_PER];
    const u32 m = msm2_stage2_rank(part_lo, part_val, b, e, loc, threadIdx.x, blockDim.x, lo, val, rank);
    __syncthreads();
    // exclusive scan of loc[0 .. gsize) in place: `per` consecutive counters per thread, wave scan of the thread sums, scan of the 16 wave sums
    const u32 per = (s.gsize + blockDim.x - 1) / blockDim.x, k0 = threadIdx.x * per;
    u32 mine = 0;
    for (u32 k = 0; k < per; k++) if (k0 + k < s.gsize) mine += loc[k0 + k];
    u32 incl = mine;
    for (int off = 1; off < 64; off <<= 1) { const u32 o = (u32)__shfl_up((int)incl, off); if ((int)(threadIdx.x & 63) >= off) incl += o; }
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    u32 before = incl - mine;
    for (u32 w = 0; w < (threadIdx.x >> 6); w++) before += wsum[w];
    for (u32 k = 0; k < per; k++) if (k0 + k < s.gsize) { const u32 c = loc[k0 + k]; loc[k0 + k] = before; before += c; }
    __syncthreads();
    msm2_stage2_place(loc, lo, val, rank, m, st_lo, st_val);
    __syncthreads();
    msm2_stage2_copy(cursor, loc, st_lo, st_val, e - b, sorted, threadIdx.x, blockDim.x);
}

// ---------------------------------------------------------------- exclusive scan of u32 (out has m+1 entries, out[m] = total)
// Every kernel of this family is a grid of SINGLE-WAVE workgroups (64 threads, scans by wave shuffles, no LDS, no barrier).  These launches
// sit between the heavy kernels of an MSM's chain, and beside them run the level-1 accumulations of the other MSMs, whose one-wave
// workgroups keep every SIMD's register file full: a four-wave workgroup needs a free slot on all four SIMDs of one CU at the same
// moment (rocprofv3: k_scan_block_sums 3.7 ms inside a proof, 12 us alone), one wave takes any slot.  Measured on the job and on the
// single proof: no difference either way (the freed slots go to the accumulations' next workgroups first); kept for the simpler kernels.
static constexpr u32 SCAN_PER_THREAD = 16, SCAN_THREADS = 64, SCAN_BLOCK = SCAN_PER_THREAD * SCAN_THREADS;
static constexpr u32 SCAN_MAX_INLINE_BLOCKS = 8 * SCAN_THREADS;   // mode 2 of k_scan_final: every workgroup scans the block sums itself, eight per lane
__device__ __forceinline__ u32 wave_inclusive_scan(u32 v) {
    for (int off = 1; off < 64; off <<= 1) { const u32 o = (u32)__shfl_up((int)v, off); if ((int)(threadIdx.x & 63) >= off) v += o; }
    return v;
}
__device__ __forceinline__ u32 wave_exclusive_scan(u32 v, u32 *total) {
    const u32 incl = wave_inclusive_scan(v);
    *total = (u32)__shfl((int)incl, 63);
    return incl - v;
}
__global__ void __launch_bounds__(64) k_scan_block_sums(const u32 *in, size_t m, u32 *block_sums) {
    size_t base = (size_t)blockIdx.x * SCAN_BLOCK + (size_t)threadIdx.x * SCAN_PER_THREAD;
    u32 s = 0;
    for (u32 k = 0; k < SCAN_PER_THREAD; k++) if (base + k < m) s += in[base + k];
    u32 total;
    wave_exclusive_scan(s, &total);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}
__global__ void __launch_bounds__(64) k_scan_of_sums(u32 *block_sums, u32 nblocks) {  // single workgroup, in place; eight sums per lane per trip
    u32 carry = 0;
    for (u32 base = 0; base < nblocks; base += 8 * 64) {
        const u32 i0 = base + threadIdx.x * 8;
        u32 v[8], mine = 0;
        for (u32 k = 0; k < 8; k++) { v[k] = i0 + k < nblocks ? block_sums[i0 + k] : 0; mine += v[k]; }
        u32 total;
        u32 ex = carry + wave_exclusive_scan(mine, &total);
        for (u32 k = 0; k < 8; k++) { if (i0 + k < nblocks) block_sums[i0 + k] = ex; ex += v[k]; }
        carry += total;
    }
    if (threadIdx.x == 0) block_sums[nblocks] = carry;
}
// mode 0: block_sums holds the exclusive scan of the block sums (+ the total at [gridDim.x]) -- after k_scan_of_sums
// mode 1: a single block: no block sums at all
// mode 2: block_sums holds the raw sums of <= SCAN_MAX_INLINE_BLOCKS blocks: every workgroup scans them itself (saves the k_scan_of_sums launch)
// The block that writes out[m] can also leave words in PINNED HOST memory (device-visible: hipHostMalloc) for the enqueueing thread: the
// total (total_host) and one more word (copy_src -> copy_host: the sort's fullest bucket).  A hipMemcpyAsync of four bytes is a blit KERNEL
// (__amd_rocclr_copyBuffer: 22 per proof, ~60 us each inside the job, every one on an MSM's chain); a store from a kernel that runs anyway is not.
__global__ void __launch_bounds__(64) k_scan_final(const u32 *in, size_t m, const u32 *block_sums, u32 *out, int mode,
                                                   u32 *total_host = nullptr, const u32 *copy_src = nullptr, u32 *copy_host = nullptr) {
    u32 offset = 0, grand = 0;
    if (mode == 0) { offset = block_sums[blockIdx.x]; grand = block_sums[gridDim.x]; }
    if (mode == 2) {   // lane t holds the sums of blocks 8 t .. 8 t + 7; this block's offset = sums of the blocks before it
        const u32 i0 = threadIdx.x * 8;
        u32 mine = 0, before_in_lane = 0;
        for (u32 k = 0; k < 8; k++) {
            const u32 v = i0 + k < gridDim.x ? block_sums[i0 + k] : 0;
            if (i0 + k < blockIdx.x) before_in_lane += v;
            mine += v;
        }
        const u32 ex = wave_exclusive_scan(mine, &grand);
        const u32 owner = blockIdx.x >> 3;   // the lane that holds this block's sum
        offset = (u32)__shfl((int)(ex + before_in_lane), (int)owner);
    }
    size_t base = (size_t)blockIdx.x * SCAN_BLOCK + (size_t)threadIdx.x * SCAN_PER_THREAD;
    u32 v[SCAN_PER_THREAD], s = 0;
    for (u32 k = 0; k < SCAN_PER_THREAD; k++) { v[k] = base + k < m ? in[base + k] : 0; s += v[k]; }
    u32 total;
    u32 ex = wave_exclusive_scan(s, &total) + offset;
    for (u32 k = 0; k < SCAN_PER_THREAD; k++) {
        if (base + k < m) out[base + k] = ex;
        ex += v[k];
    }
    if (mode == 1) grand = total;
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
        out[m] = grand;
        if (total_host) *total_host = grand;
        if (copy_host) *copy_host = *copy_src;
    }
}
// The chunk counts of the groups (msm2_chunk_count_body) and the scan of them as ONE single-wave launch (ngroups <= 4096: lane t takes the groups [t*per, (t+1)*per)):
// gstart, nchunks, cstart = exclusive scan of nchunks (+ the total at [ngroups]).  Also zeroes the two things kernels further down this
// stream add into: the fullest-bucket word (k_msm2_colsum's atomicMax) and the `nzero` block sums of the key scan (k_msm2_colsum's atomicAdd).
__global__ void __launch_bounds__(64) k_msm2_chunk_count_scan(Msm2Shape s, const u32 *S1, u32 *gstart, u32 *nchunks, u32 *cstart, u32 *max_word, u32 *key_block_sums, u32 nzero) {
    const u32 per = (s.ngroups + 63) / 64, g0 = threadIdx.x * per;
    if (threadIdx.x == 0) *max_word = 0;
    for (u32 k = threadIdx.x; k < nzero; k += 64) key_block_sums[k] = 0;
    u32 mine = 0;
    for (u32 k = 0; k < per; k++) if (g0 + k < s.ngroups) { msm2_chunk_count_body(s, S1, gstart, nchunks, g0 + k); mine += nchunks[g0 + k]; }
    u32 total;
    u32 ex = wave_exclusive_scan(mine, &total);
    for (u32 k = 0; k < per; k++) if (g0 + k < s.ngroups) { cstart[g0 + k] = ex; ex += nchunks[g0 + k]; }
    if (threadIdx.x == 0) cstart[s.ngroups] = total;
}
// prep of the next level fused with the first half of its scan: thread = SCAN_PER_THREAD consecutive keys, as in k_scan_block_sums
__global__ void __launch_bounds__(64) k_msm_prep_next_sums(u32 nkeys, const u32 *prev_items, const u32 *prev_item_start, u32 L, u32 *start, u32 *cnt,
                                                           u32 *items, u32 *block_sums) {
    size_t base = (size_t)blockIdx.x * SCAN_BLOCK + (size_t)threadIdx.x * SCAN_PER_THREAD;
    u32 s = 0;
    for (u32 k = 0; k < SCAN_PER_THREAD; k++) {
        if (base + k < nkeys) {
            msm_prep_next(prev_items, prev_item_start, L, start, cnt, items, (u32)(base + k));
            const u32 m = prev_items[base + k];
            s += m > 1 ? (m + L - 1) / L : 0;
        }
    }
    u32 total;
    wave_exclusive_scan(s, &total);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}
// the same for LEVEL 1 (k_msm_prep1's work -- decomposition of every key, infinity into the buckets of keys without entries, the finisher's
// counters zeroed -- with the block sums of the item scan): the first kernel of that scan is not launched
__global__ void __launch_bounds__(64) k_msm_prep1_sums(MsmShape s, const u32 *S, u32 L, u32 *start, u32 *cnt, u32 *items, u32 *bucket, u32 bucket_words, u32 *block_sums) {
    size_t base = (size_t)blockIdx.x * SCAN_BLOCK + (size_t)threadIdx.x * SCAN_PER_THREAD;
    u32 sum = 0;
    for (u32 k = 0; k < SCAN_PER_THREAD; k++) {
        const u32 key = (u32)(base + k);
        if (base + k < s.nkeys) {
            msm_prep_level1(s, S, L, start, cnt, items, key);
            sum += items[key];
            if (cnt[key] == 0) {
                uint4 *b = reinterpret_cast<uint4 *>(bucket + (size_t)key * bucket_words);
                for (u32 q = 0; q < bucket_words / 4; q++) b[q] = make_uint4(0, 0, 0, 0);
            }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < 16) bucket[(size_t)s.nkeys * bucket_words + threadIdx.x] = 0;
    u32 total;
    wave_exclusive_scan(sum, &total);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}
// The finisher's two lists (msm_curve_kernels.cuh k_msm_finish_keys): keys that still hold 2 .. small_max partial sums / more than
// small_max.  One atomic per wave and list; the order inside a list is whatever the atomics give (the sums do not depend on it).
// counters[0..1] are zeroed with the bucket array at the start of the accumulate stage.
__global__ void __launch_bounds__(64) k_msm_finish_list(u32 nkeys, const u32 *items, u32 small_max, u32 *list_small, u32 *list_big, u32 *counters) {
    const u32 key = blockIdx.x * 64 + threadIdx.x, lane = threadIdx.x;
    const u32 c = key < nkeys ? items[key] : 0;
    const bool small = c > 1 && c <= small_max, big = c > small_max;
    const unsigned long long ms = __ballot(small), mb = __ballot(big), below = (1ull << lane) - 1;
    u32 base_s = 0, base_b = 0;
    if (lane == 0) {
        if (ms) base_s = atomicAdd(&counters[0], (u32)__popcll(ms));
        if (mb) base_b = atomicAdd(&counters[1], (u32)__popcll(mb));
    }
    base_s = (u32)__shfl((int)base_s, 0); base_b = (u32)__shfl((int)base_b, 0);
    if (small) list_small[base_s + (u32)__popcll(ms & below)] = key;
    if (big) list_big[base_b + (u32)__popcll(mb & below)] = key;
}
// total_host / copy_src -> copy_host: words the last kernel also leaves in pinned host memory (k_scan_final), or null
static int32_t exclusive_scan(mi_ctx *ctx, hipStream_t st, const u32 *in, size_t m, u32 *out, DevBuf &tmp,
                              u32 *total_host = nullptr, const u32 *copy_src = nullptr, u32 *copy_host = nullptr) {
    u32 nblocks = (u32)((m + SCAN_BLOCK - 1) / SCAN_BLOCK);
    if (nblocks == 0) nblocks = 1;
    MI_TRY(mi_reserve(ctx, tmp, (size_t)(nblocks + 1) * 4));
    u32 *bs = (u32 *)tmp.p;
    if (nblocks == 1) {
        hipLaunchKernelGGL(k_scan_final, dim3(1), dim3(SCAN_THREADS), 0, st, in, m, bs, out, 1, total_host, copy_src, copy_host);
    } else if (nblocks <= SCAN_MAX_INLINE_BLOCKS) {
        hipLaunchKernelGGL(k_scan_block_sums, dim3(nblocks), dim3(SCAN_THREADS), 0, st, in, m, bs);
        hipLaunchKernelGGL(k_scan_final, dim3(nblocks), dim3(SCAN_THREADS), 0, st, in, m, bs, out, 2, total_host, copy_src, copy_host);
    } else {
        hipLaunchKernelGGL(k_scan_block_sums, dim3(nblocks), dim3(SCAN_THREADS), 0, st, in, m, bs);
        hipLaunchKernelGGL(k_scan_of_sums, dim3(1), dim3(SCAN_THREADS), 0, st, bs, nblocks);
        hipLaunchKernelGGL(k_scan_final, dim3(nblocks), dim3(SCAN_THREADS), 0, st, in, m, bs, out, 0, total_host, copy_src, copy_host);
    }
    MI_CHECK_HIP(ctx, hipGetLastError());
    return MI_OK;
}

// ---------------------------------------------------------------- orchestration
// An MSM is enqueued asynchronously on the stream of a *slot* (ctx->msm[i]): sort stage, accumulate
// stage, async copy of the <= 128 window sums into pinned host memory.  Nothing blocks the host until
// mi_msm_finish.  Several slots run concurrently (prove.hip puts the five MSMs on five streams so the
// latency-bound tails of one overlap the throughput-bound accumulation of another), and an accumulate
// stage may reuse another slot's sort (pk.G1.B and pk.G2.B are multiplied by the same scalars).
struct LevelArrays { u32 *start, *cnt, *items, *item_start; };

static u32 auto_c(u32 n) {
    u32 lg = 0;
    while ((2u << lg) <= n) lg++;          // floor(log2 n) for n >= 1
    int c = (int)lg - 4;
    if (c < 3) c = 3;
    if (c > 16) c = 16;
    return (u32)c;
}
static u32 msm_auto_seg(u32 nbuckets) { return nbuckets >= (1u << 16) ? 16u : nbuckets >= 256 ? 8u : 2u; }
static MsmShape slot_shape(const MsmSlot &sl) { return msm_shape(sl.n, sl.c, sl.G); }
// shape of the key space the accumulate / reduce stages see: nwin_keys windows of 2^(c-1) buckets
static MsmShape key_shape(const MsmSlot &sl) {
    MsmShape s;
    s.c = sl.c; s.nwin = sl.nwin_keys; s.nbuckets = 1u << (sl.c - 1); s.nkeys = s.nwin * s.nbuckets; s.nslices = sl.G; s.n = sl.n;
    return s;
}

// Process-wide, read by mi_msm_state_init (experiments on which chains share a hardware queue: the runtime gives a new stream the
// least-loaded of its priority class's four queues, ties to the newest -- tools/probes/stream_queue_probe.py):
//   0  slot 3 (K) created, then destroyed and pointed at slot 1's stream (rounds 4-5)          1  slot 3 never created
//   2  as 1, and slot 1 (B1 + K) on slot 0's stream (A): three wire chains instead of four
static std::atomic<int> g_stream_plan{1};
// counters the tests read to prove that an optional path really ran (names in the header)
extern "C" int32_t mi_debug_get_counter(mi_ctx *ctx, const char *name, uint64_t *out) {
    if (!ctx || !name || !out) return MI_EINVAL;
    if (!std::strcmp(name, "z_count_fused_launches")) { *out = ctx->z_count_fused_launches; return MI_OK; }
    if (!std::strcmp(name, "dense_item_sorts")) { *out = ctx->dense_item_sorts.load(); return MI_OK; }
    MI_FAIL(ctx, MI_EINVAL, std::string("unknown counter: ") + name);
}
extern "C" int32_t mi_debug_set_stream_plan(int32_t plan) {
    if (plan < 0 || plan > 2) return MI_EINVAL;
    g_stream_plan.store(plan);
    return MI_OK;
}
int32_t mi_msm_state_init(mi_ctx *ctx) {
    new (ctx->msm_knobs) MsmKnobs();
    // the c = 16 histogram / cursor image is 128 KiB of LDS (gfx950 allows 160 KiB per workgroup)
    (void)hipFuncSetAttribute((const void *)k_msm_hist, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)k_msm_scatter, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)k_msm2_hist2, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
#define MI_PART_LDS(C) (void)hipFuncSetAttribute((const void *)k_msm2_partition<C>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
    for (u32 c = 15; c <= 22; c++) { MSM2_FOR_C(c, MI_PART_LDS) }   // (15: the run-time-width instance)
#undef MI_PART_LDS
    (void)hipFuncSetAttribute((const void *)k_msm2_scatter2, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)k_msm2_scatter2_staged, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    // Stream priorities (3 levels on this device).  prove.hip runs A, B1, B2, K on slots 0..3 and Z on slot 4; computeH runs on
    // the context's own stream (high, api.hip).  With equal priorities the hardware shares the CUs evenly, all five MSMs crawl
    // along together and their latency-bound tails pile up at the end of the proof.  Z's stream at LOW priority lets the four
    // wire MSMs finish first -- their tails hide under Z's bulk -- and, with several proofs in flight, lets the next proof's
    // head run ahead of this proof's Z: 26.8 vs 24.7 proofs/s and 38.9 vs 44.6 ms single-proof latency (any assignment
    // that ranks Z below the rest measured within 1 % of that; all-equal, high or normal, did not).
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    // The runtime maps streams onto a few hardware queues in creation order, and two streams on one queue run one after the other: the
    // order in which the slots' streams are created decides WHICH two MSMs of a proof share a queue (rocprofv3: with the natural order
    // K (slot 3) queued behind B2's whole G2 chain (slot 2) and ended a single proof; every other order measured within 0.3 ms of
    // this one or worse).
    const int plan = g_stream_plan.load();
    for (int idx = 0; idx < MI_MSM_SLOTS; idx++) {
        if (plan >= 1 && idx == 3) continue;   // K's slot borrows a stream below
        if (plan >= 2 && idx == 1) continue;
        MsmSlot &sl = ctx->msm[idx];
        int pw = 0, pz = prio_lo;   // wires, Z: MI_PRIO_SOLO, MI_PRIO_POOL_SECOND
        if (ctx->prio_scheme == MI_PRIO_POOL_FIRST) { pw = prio_hi; pz = 0; }
        if (ctx->prio_scheme == MI_PRIO_POOL_REST) pw = pz = prio_lo;
        // every failure is reported: a null stream / event / host_wsum would otherwise surface much later as a memcpy into
        // null (msm_accum_enqueue).  The caller (mi_init_prio) unwinds through mi_shutdown, which frees what was created.
        MI_CHECK_HIP(ctx, hipStreamCreateWithPriority(&sl.stream, hipStreamNonBlocking, idx == 4 ? pz : pw));
    }
    // K (slot 3) runs on B1's stream (slot 1).  With a stream of its own it landed on the hardware queue of B2's stream, behind the whole
    // G2 chain, and a single proof ended with K's accumulation alone on the GPU: rocprofv3 timeline, K's level 1 starting at 27 of 31 ms.
    // Behind B1 -- the shortest of the five MSMs -- one proof is 1.0 ms shorter (30.8 against 31.9 ms) and the job unchanged; behind A
    // (whose sort K shares) 0.8 ms.  The two chains are enqueued from two host threads (mi_prove_enqueue_b_msms / _ak_msms) and so
    // interleave on the one stream: their buffers are disjoint, every wait one of them inserts also holds the other, and the event
    // pair around K's level-1 launch may bracket a kernel of B1's chain (the stats of slot 3 are then an upper bound).
    if (ctx->msm[3].stream) (void)hipStreamDestroy(ctx->msm[3].stream);
    if (plan >= 2) ctx->msm[1].stream = ctx->msm[0].stream;
    ctx->msm[3].stream = ctx->msm[1].stream;
    for (auto &sl : ctx->msm) {
        for (auto &e : sl.ev) MI_CHECK_HIP(ctx, hipEventCreate(&e));
        MI_CHECK_HIP(ctx, hipHostMalloc(&sl.host_wsum, 128 * 256 + 64));
    }
    return MI_OK;
}
void mi_msm_state_free(mi_ctx *ctx) {
    for (int i = 0; i < MI_MSM_SLOTS; i++) for (int j = i + 1; j < MI_MSM_SLOTS; j++) if (ctx->msm[j].stream == ctx->msm[i].stream) ctx->msm[j].stream = nullptr;   // aliases
    for (auto &sl : ctx->msm) {
        if (sl.stream) { (void)hipStreamSynchronize(sl.stream); (void)hipStreamDestroy(sl.stream); }
        for (auto &e : sl.ev) if (e) (void)hipEventDestroy(e);
        if (sl.host_wsum) (void)hipHostFree(sl.host_wsum);
        for (auto &b : sl.buf) if (b.p) (void)hipFree(b.p);
    }
}

// slot buffers
enum { B_DIGITS, B_H, B_S, B_SORTED, B_LEVELS, B_PART0, B_PART1, B_BUCKET, B_SCAN, B_WIN, B_PVAL, B_C1, B_CHUNKS, B_ITEMTAB, B_MAX, B_BA_NODES, B_BA_PREFIX, B_BA_TOT, B_COUNT_ };
static_assert(B_COUNT_ <= sizeof(MsmSlot::buf) / sizeof(DevBuf), "MsmSlot::buf is too small");

// The fullest bucket of a sort: per-key totals -> one word (atomicMax), copied to pinned host memory behind the slot's ev[6].  The item
// machinery needs ceil(log_L(fullest bucket)) levels; the worst case (every entry in one bucket) says 9 at N = 2^23 where uniform
// scalars need 3 and the WHIR mix 7, and every unneeded level is three launches of empty kernels on the MSM's tail.  The host waits for
// the word on the thread that enqueues the accumulation (a helper thread for the wire MSMs) while the rest of the sort still runs.
__global__ void __launch_bounds__(256) k_max_u32(const u32 *v, u32 n, u32 *out) {
    u32 m = 0;
    for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) m = v[i] > m ? v[i] : m;
    for (int off = 32; off > 0; off >>= 1) { const u32 o = (u32)__shfl_xor((int)m, off); m = o > m ? o : m; }
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}
// totals == nullptr: the word at sl.buf[B_MAX] has been computed already (k_msm2_colsum); only the copy and the event are enqueued
// Then the scan of the per-key totals -> keystart, whose last kernel ALSO leaves that word (host_wsum + 128*256 + 32) and the number of
// sorted entries (host_wsum + 128*256) in the slot's pinned host memory: no copy launches (k_scan_final); ev[6] follows the scan.
// sums_ready: the scan's block sums are in sl.buf[B_SCAN] already (k_msm2_colsum added them up): only the scan's last kernel is launched.
static int32_t fetch_max_and_scan_keys(mi_ctx *ctx, MsmSlot &sl, const u32 *totals, u32 nkeys, u32 *keystart, bool compute_max, bool sums_ready = false) {
    MI_TRY(mi_reserve(ctx, sl.buf[B_MAX], 64));
    u32 *dmax = (u32 *)sl.buf[B_MAX].p;
    hipStream_t st = sl.stream;
    if (compute_max) {
        MI_CHECK_HIP(ctx, hipMemsetAsync(dmax, 0, 4, st));
        unsigned grid = (nkeys + 255) / 256;
        if (grid > 1024) grid = 1024;
        hipLaunchKernelGGL(k_max_u32, dim3(grid), dim3(256), 0, st, totals, nkeys, dmax);
        MI_CHECK_HIP(ctx, hipGetLastError());
    }
    char *host = (char *)sl.host_wsum + 128 * 256;
    if (sums_ready) {
        const u32 nblocks = (nkeys + SCAN_BLOCK - 1) / SCAN_BLOCK;
        hipLaunchKernelGGL(k_scan_final, dim3(nblocks), dim3(SCAN_THREADS), 0, st, totals, (size_t)nkeys, (const u32 *)sl.buf[B_SCAN].p, keystart, 2, (u32 *)host, (const u32 *)dmax,
                           (u32 *)(host + 32));
        MI_CHECK_HIP(ctx, hipGetLastError());
    } else {
        MI_TRY(exclusive_scan(ctx, st, totals, nkeys, keystart, sl.buf[B_SCAN], (u32 *)host, dmax, (u32 *)(host + 32)));
    }
    MI_CHECK_HIP(ctx, hipEventRecord(sl.ev[6], st));
    sl.max_pending = true;
    return MI_OK;
}

// Runs levels of the item machinery over `nkeys` keys whose level-0 decomposition (start/cnt/items) is
// already in cur.  Level 0 reads (pts, sorted) when pts != null, else partial_first.
static int32_t run_levels(mi_ctx *ctx, const MsmCurveOps &ops, MsmSlot &sl, u32 nkeys, LevelArrays cur, LevelArrays nxt, u64 first_items_bound,
                          u64 max_count, u32 L_first, u32 L_next, const void *pts, const u32 *sorted, const void *partial_first,
                          void *final_out, bool time_first, bool rprime = false, bool first_sums_ready = false) {
    hipStream_t st = sl.stream;
    const void *pin = partial_first;
    u64 items_bound = first_items_bound;
    u64 m = max_count;  // bound on entries of the largest key at this level
    u32 L = L_first;
    const u32 scan_blocks = (nkeys + SCAN_BLOCK - 1) / SCAN_BLOCK;
    // partial sums between the levels in the packed R' form (curve29.cuh) when level 1 runs in 29-bit limbs and the curve has the
    // matching upper-level kernel
    const bool rp_partials = pts && rprime && ops.accum_affine_rp && ops.accum_xyzz_rp && !knobs_of(ctx)->std_partials;
    if (first_sums_ready) {   // (k_msm_prep1_sums left the block sums of cur.items in B_SCAN)
        hipLaunchKernelGGL(k_scan_final, dim3(scan_blocks), dim3(SCAN_THREADS), 0, st, cur.items, (size_t)nkeys, (const u32 *)sl.buf[B_SCAN].p, cur.item_start, 2, nullptr, nullptr, nullptr);
        MI_CHECK_HIP(ctx, hipGetLastError());
    } else {
        MI_TRY(exclusive_scan(ctx, st, cur.items, nkeys, cur.item_start, sl.buf[B_SCAN]));
    }
    for (u32 level = 0;; level++) {
        DevBuf &pout_buf = sl.buf[(level & 1) ? B_PART1 : B_PART0];
        MI_TRY(mi_reserve(ctx, pout_buf, (items_bound + 1) * ops.xyzz_bytes));
        void *pout = pout_buf.p;
        // bounded grid, grid-stride inside: at most 128 single-wave workgroups per CU, i.e. a wave of the level-1 kernel lives for a few
        // items (~0.5 ms), not for the whole launch.  A fully persistent grid (32 per CU = every wave slot) made the small kernels of the
        // other MSM streams wait for the end of the launch: 128..4096 per CU measured +1.5 % proofs/s and -1 ms latency over 32.
        // Levels that turn out to be (nearly) empty -- the bound is a worst case -- still cost microseconds, not a full dispatch.
        u32 grid = (u32)((items_bound + 63) / 64);
        u32 grid_cap = (u32)ctx->cu_count * 128;
        {   // knobs g1_grid_per_cu / g2_grid_per_cu: resident-grid caps per CU for the level-1 kernels
            const u32 capx = ops.xyzz_bytes == 256 ? knobs_of(ctx)->g2_grid_per_cu : knobs_of(ctx)->g1_grid_per_cu;
            if (level == 0 && capx > 0) grid_cap = (u32)ctx->cu_count * capx;
        }
        if (grid > grid_cap) grid = grid_cap;
        if (grid == 0) grid = 1;
        // the timed span (mi_stats.g1_accum_kernel_ms) brackets the accumulate kernel alone: on the 29-bit path the launcher records the
        // opening event AFTER its item-table kernel (0.1 ms alone, up to 0.5 ms waiting for CUs with three proofs in flight)
        const bool rp_path = level == 0 && pts && rprime && ops.accum_affine_rp;
        if (time_first && level == 0 && !rp_path) MI_CHECK_HIP(ctx, hipEventRecord(sl.ev[1], st));
        const u32 ba_rounds = knobs_of(ctx)->ba_rounds;
        // batch-affine rounds (msm_ba_g1.cuh) where the buckets hold a few items each (>= 32 entries on average) and the scratch fits
        const u32 ba_waves = (u32)ctx->cu_count * 16;
        const size_t ba_tot = msm_ba_scratch_bytes(items_bound + 1, ba_waves);
        const bool use_ba = rp_path && ba_rounds && ops.accum_affine_ba && L == 16 && first_items_bound >= (u64)nkeys * 3 &&
                            mi_try_reserve(sl.buf[B_BA_NODES], (items_bound + 1) * 512) && mi_try_reserve(sl.buf[B_BA_PREFIX], (items_bound + 1) * 256) &&
                            mi_try_reserve(sl.buf[B_BA_TOT], 2 * ba_tot);
        if (use_ba) {
            MI_TRY(mi_reserve(ctx, sl.buf[B_ITEMTAB], (items_bound + 1) * 16));
            ops.accum_affine_ba(st, (u32)ctx->cu_count * 64, pts, sorted, cur.start, cur.cnt, cur.items, cur.item_start, nkeys, final_out, pout, sl.buf[B_ITEMTAB].p,
                                rp_partials ? 1u : 0u, ba_rounds, items_bound + 1, ba_waves, sl.buf[B_BA_NODES].p, sl.buf[B_BA_PREFIX].p, sl.buf[B_BA_TOT].p,
                                (char *)sl.buf[B_BA_TOT].p + ba_tot, time_first ? sl.ev[1] : nullptr);
        } else if (rp_path) {
            MI_TRY(mi_reserve(ctx, sl.buf[B_ITEMTAB], (items_bound + 1) * 16));
            // which build of the G1 level-1 kernel: three waves per SIMD (168 VGPRs: a SIMD's register file is full, and a freed wave slot
            // is too small for any 256-VGPR G2 workgroup, which then waits for the END of this launch) or two (196: one freed slot admits one)
            const MsmKnobs *kn = knobs_of(ctx);
            const bool g2 = ops.xyzz_bytes == 256;
            const bool two = !g2 && (kn->l1_waves == 2 || (kn->z_waves == 2 && &sl == &ctx->msm[4]));
            const u32 wg = g2 ? kn->g2_wg : kn->l1_wg, wg_log = wg == 4 ? 2u : wg == 2 ? 1u : 0u;   // waves per workgroup
            // (the level-1 launches on lowest-priority streams of their own -- "the accumulation is what fills the GPU, everything else is
            //  dispatched ahead of it" -- measured -6 % proofs/s, profiles/r05_ab_evidence.txt: removed)
            ops.accum_affine_rp(st, grid, pts, sorted, cur.start, cur.cnt, cur.items, cur.item_start, nkeys, L, final_out, pout, sl.buf[B_ITEMTAB].p,
                                (rp_partials ? 1u : 0u) | (two ? 2u : 0u) | (wg_log << 2), time_first ? sl.ev[1] : nullptr);
        } else if (level == 0 && pts) ops.accum_affine(st, grid, pts, sorted, cur.start, cur.cnt, cur.items, cur.item_start, nkeys, L, final_out, pout);
        else if (rp_partials) ops.accum_xyzz_rp(st, grid, pin, cur.start, cur.cnt, cur.items, cur.item_start, nkeys, L, final_out, pout);
        else ops.accum_xyzz(st, grid, pin, cur.start, cur.cnt, cur.items, cur.item_start, nkeys, L, final_out, pout);
        MI_CHECK_HIP(ctx, hipGetLastError());
        if (time_first && level == 0) MI_CHECK_HIP(ctx, hipEventRecord(sl.ev[2], st));
        u64 m_next = (m + L - 1) / L;  // entries of the largest key at the next level
        if (m_next <= 1) break;
        // the finisher: no key holds more than finish_max partial sums -> one list launch + one launch end the machinery (the lists live in
        // the next level's start / cnt arrays, free from here on; the counters behind the bucket array, zeroed with it)
        {
            const MsmKnobs *kn = knobs_of(ctx);
            const u64 fin_max = kn->finisher_max ? kn->finisher_max : ops.finish_max;
            if (kn->finisher && ops.finish_keys && m_next <= fin_max && level >= kn->finisher_min_level) {
                u32 *counters = (u32 *)((char *)final_out + (size_t)nkeys * ops.xyzz_bytes);
                hipLaunchKernelGGL(k_msm_finish_list, dim3((nkeys + 63) / 64), dim3(64), 0, st, nkeys, cur.items, MSM_FIN_SMALL, nxt.start, nxt.cnt, counters);
                const u32 T = ops.finish_T;
                u32 nb_small = (nkeys + T - 1) / T, nb_big = nkeys;
                const u32 cap_small = (u32)ctx->cu_count * 8, cap_big = (u32)ctx->cu_count * 4;
                if (nb_small > cap_small) nb_small = cap_small;
                if (nb_big > cap_big) nb_big = cap_big;
                ops.finish_keys(st, nb_small, nb_big, pout, nxt.start, nxt.cnt, counters, cur.item_start, cur.items, final_out, rp_partials ? 1u : 0u);
                MI_CHECK_HIP(ctx, hipGetLastError());
                break;
            }
        }
        // next level: (start, cnt, items) of the keys that go on, and the exclusive scan of their items
        if (scan_blocks <= SCAN_MAX_INLINE_BLOCKS) {   // prep fused with the block sums, the scan of the sums fused with the final pass: two launches
            MI_TRY(mi_reserve(ctx, sl.buf[B_SCAN], (size_t)(scan_blocks + 1) * 4));
            u32 *bs = (u32 *)sl.buf[B_SCAN].p;
            hipLaunchKernelGGL(k_msm_prep_next_sums, dim3(scan_blocks), dim3(SCAN_THREADS), 0, st, nkeys, cur.items, cur.item_start, L_next, nxt.start, nxt.cnt,
                               nxt.items, bs);
            hipLaunchKernelGGL(k_scan_final, dim3(scan_blocks), dim3(SCAN_THREADS), 0, st, nxt.items, (size_t)nkeys, bs, nxt.item_start, 2, nullptr, nullptr, nullptr);
            MI_CHECK_HIP(ctx, hipGetLastError());
        } else {
            hipLaunchKernelGGL(k_msm_prep_next, dim3((nkeys + 63) / 64), dim3(64), 0, st, nkeys, cur.items, cur.item_start, L_next, nxt.start, nxt.cnt, nxt.items);
            MI_CHECK_HIP(ctx, hipGetLastError());
            MI_TRY(exclusive_scan(ctx, st, nxt.items, nkeys, nxt.item_start, sl.buf[B_SCAN]));
        }
        // items at the next level: every continuing key has >= 2 entries, so items <= entries/L + keys
        u64 nb = items_bound / L_next + (items_bound < nkeys ? items_bound : nkeys) + 1;
        items_bound = nb < items_bound ? nb : items_bound;
        m = m_next; L = L_next; pin = pout;
        LevelArrays t = cur; cur = nxt; nxt = t;
    }
    return MI_OK;
}

// sort stage on slot sl: digits + counting sort of (key -> point index | sign).  Records sl.ev[0].
static int32_t msm2_sort_enqueue(mi_ctx *ctx, MsmSlot &sl, const Fr *scalars, u32 n, u32 flags, u32 c, bool wkeys = false, bool exact = false);
static int32_t msm_sort_enqueue(mi_ctx *ctx, MsmSlot &sl, const Fr *scalars, u32 n, u32 flags, u32 generic_c, bool exact) {
    MsmKnobs *kn = knobs_of(ctx);
    sl.n = n;
    sl.c = kn->c ? kn->c : (generic_c >= 2 && generic_c <= 16 ? generic_c : auto_c(n));
    // Large generic MSMs (c = 16: 16 windows x 2^15 buckets = 2^19 keys, as many as a fixed-base c = 20 sort) borrow the
    // fixed-base path's two-pass sort with the window folded into the key.  The one-pass scatter below writes every 4-byte
    // entry as a partial line of its own (PMC round 1: 4.1 GB written for 0.5 GB of entries at 2^23 pairs); the two-pass
    // sort stages runs through LDS and walks its chunks XCD by XCD (1.5x).  Same entries, same order inside a key up to the
    // order of LDS atomics -- which the sums do not depend on.
    if (sl.c == 16 && !kn->one_pass_sort && n >= (1u << 18) && (u64)n * 16 < ((u64)1 << 31)) return msm2_sort_enqueue(ctx, sl, scalars, n, flags, 16, true, exact);
    sl.G = kn->G ? kn->G : (n / 8192 > 64 ? 64 : (n / 8192 ? n / 8192 : 1));
    const MsmShape s = slot_shape(sl);
    sl.nwin_keys = sl.nwin_digits = s.nwin;
    const u64 T_bound = (u64)s.nwin * n;
    sl.entries_cap = T_bound;
    if (s.nwin > 128) MI_FAIL(ctx, MI_EINVAL, "msm: too many windows");
    MI_TRY(mi_reserve(ctx, sl.buf[B_DIGITS], T_bound * 2 + 64));
    MI_TRY(mi_reserve(ctx, sl.buf[B_H], ((size_t)s.nkeys * s.nslices + 1) * 4));
    MI_TRY(mi_reserve(ctx, sl.buf[B_S], ((size_t)s.nkeys * 2 + 2) * 4));
    MI_TRY(mi_reserve(ctx, sl.buf[B_SORTED], (T_bound + 1) * 4));
    int16_t *digits = (int16_t *)sl.buf[B_DIGITS].p;
    u32 *H = (u32 *)sl.buf[B_H].p, *keystart = (u32 *)sl.buf[B_S].p, *total = keystart + s.nkeys + 1, *sorted = (u32 *)sl.buf[B_SORTED].p;
    hipStream_t st = sl.stream;
    hipLaunchKernelGGL(k_msm_digits, dim3((n + 255) / 256), dim3(256), 0, st, s, scalars, (flags & MI_MSM_SCALARS_CANONICAL) ? 0 : 1, digits);
    const size_t lds_bytes = (size_t)s.nbuckets * 4;
    hipLaunchKernelGGL(k_msm_hist, dim3(s.nslices, s.nwin), dim3(1024), lds_bytes, st, s, digits, H);
    MI_CHECK_HIP(ctx, hipGetLastError());
    hipLaunchKernelGGL(k_msm_colsum, dim3((s.nkeys + 255) / 256), dim3(256), 0, st, s, H, total);
    MI_CHECK_HIP(ctx, hipGetLastError());
    MI_TRY(fetch_max_and_scan_keys(ctx, sl, total, s.nkeys, keystart, true));
    hipLaunchKernelGGL(k_msm_scatter, dim3(s.nslices, s.nwin), dim3(1024), lds_bytes, st, s, digits, keystart, H, sorted);
    MI_CHECK_HIP(ctx, hipGetLastError());
    MI_CHECK_HIP(ctx, hipEventRecord(sl.ev[0], st));
    return MI_OK;
}

// fixed-base sort stage: entries of ALL windows keyed by one bucket set of 2^(c-1), two-pass sort.  Records sl.ev[0].
// the shape of the fixed-base sort of n scalars with c-bit windows under the context's knobs (one place: the sort and the count hook must agree)
static Msm2Shape msm2_plan_shape(const MsmKnobs *kn, u32 n, u32 c, bool wkeys) {
    const u32 G = n ? (n + MSM2_SLICE - 1) / MSM2_SLICE : 1;   // pass-1 slices
    const u32 chunk = kn->chunk ? kn->chunk : 8192;    // (gbits, chunk) sweep at 2^23 pairs, c = 20: tools/fixed_probe.py
    u32 gbits = kn->gbits ? kn->gbits : 11;
    if (gbits > 15) gbits = 15;
    const u32 keys_total = (wkeys ? (256 + c - 1) / c : 1u) << (c - 1);
    while ((keys_total >> gbits) > MSM2_MAX_GROUPS) gbits++;
    return msm2_shape(n, c, G, chunk, gbits, wkeys ? 1u : 0u);
}
int32_t mi_msm_z_count_arm(mi_ctx *ctx, int slot, size_t n, uint32_t c) {
    static_assert(sizeof(Msm2Shape) <= sizeof(ctx->zhook.shape), "Msm2Shape travels in ctx->zhook.shape");
    ctx->zhook.armed = ctx->zhook.done = false; ctx->zhook.h = nullptr;
    const MsmKnobs *kn = knobs_of(ctx);
    if (!kn->z_count_fused || slot != MI_ZHOOK_SLOT || c < 17 || c > 22 || n < MSM2_SLICE || n > ((size_t)1 << 27)) return MI_OK;
    const Msm2Shape s = msm2_plan_shape(kn, (u32)n, c, false);
    if ((u64)s.nwin * n >= ((u64)1 << 31) || (n + s.nslices - 1) / s.nslices != MSM2_SLICE) return MI_OK;
    MsmSlot &sl = ctx->msm[slot];
    MI_TRY(mi_reserve(ctx, sl.buf[B_C1], ((size_t)s.ngroups * s.nslices + 1) * 4 * 2));
    std::memcpy(ctx->zhook.shape, &s, sizeof(s));
    ctx->zhook.slot = slot; ctx->zhook.n = (u32)n; ctx->zhook.c = c; ctx->zhook.C1 = (u32 *)sl.buf[B_C1].p;
    ctx->zhook.armed = true;
    return MI_OK;
}
static int32_t msm2_sort_enqueue(mi_ctx *ctx, MsmSlot &sl, const Fr *scalars, u32 n, u32 flags, u32 c, bool wkeys, bool exact) {
    MsmKnobs *kn = knobs_of(ctx);
    if (!wkeys && (c < 17 || c > 22)) MI_FAIL(ctx, MI_EINVAL, "fixed-base msm: window bits must be 17..22");
    const Msm2Shape s = msm2_plan_shape(kn, n, c, wkeys);
    const u32 G = s.nslices, chunk = s.chunk;
    sl.n = n; sl.c = c; sl.G = G; sl.nwin_keys = wkeys ? s.nwin : 1; sl.nwin_digits = s.nwin;
    const u64 T_bound = (u64)s.nwin * n;
    if (T_bound >= ((u64)1 << 31)) MI_FAIL(ctx, MI_EINVAL, "fixed-base msm: windows * n must stay below 2^31");
    MI_TRY(mi_reserve(ctx, sl.buf[B_C1], ((size_t)s.ngroups * G + 1) * 4 * 2));
    MI_TRY(mi_reserve(ctx, sl.buf[B_CHUNKS], ((size_t)s.ngroups + 1) * 4 * 3));
    MI_TRY(mi_reserve(ctx, sl.buf[B_S], ((size_t)s.nkeys * 2 + 2) * 4));
    u32 *C1 = (u32 *)sl.buf[B_C1].p, *S1 = C1 + (size_t)s.ngroups * G + 1;
    const int mont = (flags & MI_MSM_SCALARS_CANONICAL) ? 0 : 1;
    hipStream_t st = sl.stream;
    // slices per counting workgroup: a group's counters leave as ONE segment of per x 4 B (per = 8: 32-B segments, 255 MB written for a
    // 16 MB matrix at N = 2^23 by the PMC counters; 32: whole 128-B lines)
    u32 per = kn->count_per >= 1 && kn->count_per <= 64 ? kn->count_per : 32u;
    while (per > 1 && per * s.ngroups > 8192) per >>= 1;
    // computeH's last launch may have counted these very scalars already (ctx->zhook, armed by prove.hip for this slot, shape and matrix)
    bool counted = false;
    if (&sl == &ctx->msm[MI_ZHOOK_SLOT]) {   // (the other slots' sorts run on helper threads and never touch the hook)
        counted = ctx->zhook.done && ctx->zhook.slot == MI_ZHOOK_SLOT && ctx->zhook.n == n && ctx->zhook.c == c && ctx->zhook.C1 == C1 && ctx->zhook.h == (const void *)scalars && mont && !wkeys;
        ctx->zhook.done = false;
    }
    if (!counted) {
#define MI_LAUNCH_COUNT(C) hipLaunchKernelGGL(k_msm2_count<C>, dim3((G + per - 1) / per), dim3(512), 0, st, s, scalars, mont, per, C1)
    MSM2_FOR_C(s.c, MI_LAUNCH_COUNT)
#undef MI_LAUNCH_COUNT
    }
    MI_CHECK_HIP(ctx, hipGetLastError());
    u32 *host_total = (u32 *)((char *)sl.host_wsum + 128 * 256 + 16);
    MI_TRY(exclusive_scan(ctx, st, C1, (size_t)s.ngroups * G, S1, sl.buf[B_SCAN], exact ? host_total : nullptr));
    // Entry-indexed workspaces (two partition arrays, chunk histograms, sorted entries; later the per-item partial sums) take
    // ~19 B per entry.  The bound nwin * n is tight for uniform scalars (the h coefficients of the Z MSM) but 3x too large
    // for wire values (45 % of them in {0, 1}, 25 % bytes: 4..5 non-zero digits of 14).  `exact`: the count pass
    // above has the true number -- fetch it (one stream synchronisation of THIS slot's stream: everything else the caller
    // enqueued keeps the GPU busy meanwhile) and size by it.  N = 2^26: 80 -> 35 GB of workspaces per context.
    u64 T = T_bound;
    if (exact) {
        MI_CHECK_HIP(ctx, hipStreamSynchronize(st));   // (the scan's last kernel stored the total in pinned host memory)
        T = *host_total;
        if (T > T_bound) MI_FAIL(ctx, MI_EHIP, "msm: counted more entries than windows * n");
    }
    sl.entries_cap = T;
    const u32 chunks_bound = (u32)(T / chunk) + s.ngroups + 1;
    MI_TRY(mi_reserve(ctx, sl.buf[B_DIGITS], T * 2 + 64));                  // part_lo (u16)
    MI_TRY(mi_reserve(ctx, sl.buf[B_PVAL], (T + 1) * 4));                   // part_val
    MI_TRY(mi_reserve(ctx, sl.buf[B_H], (size_t)chunks_bound * s.gsize * 4));
    MI_TRY(mi_reserve(ctx, sl.buf[B_SORTED], (T + 1) * 4));
    uint16_t *part_lo = (uint16_t *)sl.buf[B_DIGITS].p;
    u32 *part_val = (u32 *)sl.buf[B_PVAL].p, *gstart = (u32 *)sl.buf[B_CHUNKS].p, *cstart = gstart + s.ngroups + 1, *nchunks = cstart + s.ngroups + 1;
    u32 *H2 = (u32 *)sl.buf[B_H].p, *keystart = (u32 *)sl.buf[B_S].p, *total = keystart + s.nkeys + 1, *sorted = (u32 *)sl.buf[B_SORTED].p;
    const u32 cap = ((n + G - 1) / G) * s.nwin;   // entries of one slice at most
    const size_t part_lds = ((size_t)3 * s.ngroups + 1 + 256 + cap) * 4 + (size_t)cap * 2 * 2;   // ... + stage_val | stage_lo, stage_grp (u16)
    if (part_lds > 160 * 1024) MI_FAIL(ctx, MI_EINVAL, "fixed-base msm: pass-1 slice does not fit in LDS");
#define MI_LAUNCH_PART(C) hipLaunchKernelGGL(k_msm2_partition<C>, dim3(((G + 7) / 8) * 8), dim3(MSM2_PART_THREADS), part_lds, st, s, scalars, mont, S1, cap, part_lo, part_val)
    MSM2_FOR_C(s.c, MI_LAUNCH_PART)
#undef MI_LAUNCH_PART
    MI_TRY(mi_reserve(ctx, sl.buf[B_MAX], 64));
    // three launches less on this chain than rounds 1-4 had: the chunk counts and their scan are one single-wave kernel, and the key scan's
    // block sums are added up by k_msm2_colsum (up to 2^19 keys: 512 block sums, what k_scan_final's inline mode takes)
    const u32 key_blocks = (s.nkeys + SCAN_BLOCK - 1) / SCAN_BLOCK;
    const bool fuse_keys = key_blocks > 1 && key_blocks <= SCAN_MAX_INLINE_BLOCKS;
    MI_TRY(mi_reserve(ctx, sl.buf[B_SCAN], (size_t)(key_blocks + 1) * 4));
    u32 *key_bs = (u32 *)sl.buf[B_SCAN].p;
    hipLaunchKernelGGL(k_msm2_chunk_count_scan, dim3(1), dim3(64), 0, st, s, S1, gstart, nchunks, cstart, (u32 *)sl.buf[B_MAX].p, key_bs, fuse_keys ? key_blocks : 0u);
    MI_CHECK_HIP(ctx, hipGetLastError());
    hipLaunchKernelGGL(k_msm2_hist2, dim3(chunks_bound), dim3(1024), s.gsize * 4, st, s, gstart, cstart, part_lo, H2);
    hipLaunchKernelGGL(k_msm2_colsum, dim3((s.nkeys + 63) / 64), dim3(64), 0, st, s, cstart, H2, total, (u32 *)sl.buf[B_MAX].p, fuse_keys ? key_bs : nullptr);
    MI_CHECK_HIP(ctx, hipGetLastError());
    MI_TRY(fetch_max_and_scan_keys(ctx, sl, total, s.nkeys, keystart, false, fuse_keys));   // (the fullest bucket: k_msm2_colsum left it in B_MAX)
    // staged (destination-order) scatter where a chunk fits eight entries per thread and two workgroups still share a CU's LDS
    const bool plain_scatter = kn->plain_scatter != 0;   // (tests and A/Bs: mi_debug_set_knob "plain_scatter")
    const size_t staged_lds = ((size_t)2 * s.gsize + 16 + chunk) * 4 + (size_t)chunk * 2;
    if (!plain_scatter && chunk <= MSM2_STAGE_PER * 1024 && staged_lds <= 80 * 1024)
        hipLaunchKernelGGL(k_msm2_scatter2_staged, dim3(chunks_bound + 8), dim3(1024), staged_lds, st, s, gstart, cstart, keystart, H2, part_lo, part_val, sorted);
    else
        hipLaunchKernelGGL(k_msm2_scatter2, dim3(chunks_bound + 8), dim3(1024), s.gsize * 4, st, s, gstart, cstart, keystart, H2, part_lo, part_val, sorted);
    MI_CHECK_HIP(ctx, hipGetLastError());
    MI_CHECK_HIP(ctx, hipEventRecord(sl.ev[0], st));
    return MI_OK;
}

// accumulate stage on slot acc, reading the sort of slot srt (may be the same slot)
static int32_t msm_tail_enqueue(mi_ctx *ctx, const MsmCurveOps &ops, MsmSlot &acc);
static int32_t msm_accum_enqueue(mi_ctx *ctx, const MsmCurveOps &ops, MsmSlot &srt, MsmSlot &acc, const void *pts, bool timed, bool defer_reduce, bool rprime) {
    MsmKnobs *kn = knobs_of(ctx);
    acc.n = srt.n; acc.c = srt.c; acc.G = srt.G; acc.nwin_keys = srt.nwin_keys; acc.nwin_digits = srt.nwin_digits;
    const MsmShape s = key_shape(srt);
    const u32 n = s.n;
    u32 L1 = kn->L1 ? kn->L1 : 16;
    const u32 L2 = kn->L2 ? kn->L2 : 8;   // tools/tune.py sweep, N = 2^23
    // buckets per bucket-reduce thread: a thread spends 2 seg additions on its segment and ~19 addition-equivalents on the multiple of its
    // running sum by the segment's base, so the one big bucket set of a fixed-base MSM (2^16 .. 2^21 buckets) wants the longer segment
    // (same-box A/B at N = 2^23: 16 against 8 +0.6 % proofs/s in 7 of 9 pairs; 32 equal, 64 and 4 slower)
    const u32 seg = kn->seg ? kn->seg : msm_auto_seg(s.nbuckets);
    const u64 T_bound = srt.entries_cap;   // nwin * n, or the counted number of entries (msm2_sort_enqueue, exact)
    hipStream_t st = acc.stream;
    if (&srt != &acc) MI_CHECK_HIP(ctx, hipStreamWaitEvent(st, srt.ev[0], 0));
    MI_TRY(mi_reserve(ctx, acc.buf[B_LEVELS], ((size_t)s.nkeys + 1) * 4 * 8));
    MI_TRY(mi_reserve(ctx, acc.buf[B_BUCKET], (size_t)s.nkeys * ops.xyzz_bytes + 64));   // + the finisher's two list counters
    const u32 *S = (const u32 *)srt.buf[B_S].p, *sorted = (const u32 *)srt.buf[B_SORTED].p;
    u32 *la = (u32 *)acc.buf[B_LEVELS].p;
    const size_t stride = (size_t)s.nkeys + 1;
    LevelArrays A{la, la + stride, la + 2 * stride, la + 3 * stride}, B{la + 4 * stride, la + 5 * stride, la + 6 * stride, la + 7 * stride};
    void *bucket = acc.buf[B_BUCKET].p;
    // largest possible bucket: one entry per scalar and window of the key space it collects -- or, fetched from this very sort, the
    // fullest bucket there is (k_max_u32 above: the wait is on the enqueueing thread and ends before the sort does)
    u64 max_count = (u64)(srt.nwin_digits / srt.nwin_keys) * n;
    if (srt.max_pending) {
        MI_CHECK_HIP(ctx, hipEventSynchronize(srt.ev[6]));
        srt.max_key_count = *(const u32 *)((const char *)srt.host_wsum + 128 * 256 + 32);
        srt.max_pending = false;
    }
    if (!kn->bound_levels && srt.max_key_count && srt.max_key_count < max_count) max_count = srt.max_key_count;
    // A FLAT sort (uniform scalars: the h coefficients of prove's Z MSM, ~208 entries in every one of 2^19 buckets at N = 2^23): with 16
    // entries per item a bucket leaves level 1 as 13 partial sums -- two items of level 2, then a third level.  An item size of average /
    // L2^k (26 there) makes the levels above full L2-ary trees: 8 partial sums, ONE item, no third level (same-process A/B: +0.4..0.6 % /
    // +0.7 %, 11 and 12 of 12 rounds).  The fullest bucket and the entry count are in host memory by now (the key scan stored them).
    bool flat = false;
    if (kn->flat_L1 != 1 && pts && srt.max_key_count) {   // (a flat sort's own size wins over the plan's L1)
        const u64 entries = *(const u32 *)((const char *)srt.host_wsum + 128 * 256);
        const u64 avg = entries / (s.nkeys ? s.nkeys : 1);
        if (avg >= 64 && (u64)srt.max_key_count <= 2 * avg) {
            if (kn->flat_L1 >= 4) { L1 = kn->flat_L1; flat = true; }
            else {
                u64 t = avg;
                while (t > 32) t = (t + L2 - 1) / L2;
                if (t >= 17) { L1 = (u32)t; flat = true; }   // (flat = the rule chose: a flat sort it has no size for falls through to the dense rule)
            }
        }
    }
    // A DENSE sort that is not flat (the wire values of a witness that is mostly full-width field elements -- what the reference's circuit
    // implies, profiles/r06_wire_census.txt -- with its bytes and bits piled into a few buckets of window 0): >= half of the n * nwin digits are
    // entries.  Items of 32 halve the partial sums the dearer upper levels add up; it triggers only where it pays (same-process A/B of the rule: census mix +0.6 % / +1.5 %, uniform +0.2 % / +0.8 %,
    // 13 of 14 rounds; BASELINE mix, ~0.3 of the digits non-zero: items of 32 measured -0.5 % in r5, the rule leaves it at 16).  The count is the sort's own, per call.
    if (!flat && !kn->L1 && kn->dense_L1 != 1 && pts && srt.max_key_count && srt.nwin_keys == 1) {
        const u64 entries = *(const u32 *)((const char *)srt.host_wsum + 128 * 256);
        if (entries >= ((u64)1 << 20) && 2 * entries >= (u64)n * srt.nwin_digits) { L1 = kn->dense_L1 >= 4 ? kn->dense_L1 : 32; ctx->dense_item_sorts++; }
    }
    // level-1 decomposition of every key (also: empty keys' buckets = infinity, finisher counters = 0) -- with the block sums of the item
    // scan whenever that scan's inline mode takes them (up to 2^19 keys): one launch less in front of every level-1 accumulation
    const u32 prep_blocks = (s.nkeys + SCAN_BLOCK - 1) / SCAN_BLOCK;
    const bool prep_sums = prep_blocks <= SCAN_MAX_INLINE_BLOCKS;
    if (prep_sums) {
        MI_TRY(mi_reserve(ctx, acc.buf[B_SCAN], (size_t)(prep_blocks + 1) * 4));
        hipLaunchKernelGGL(k_msm_prep1_sums, dim3(prep_blocks), dim3(SCAN_THREADS), 0, st, s, S, L1, A.start, A.cnt, A.items, (u32 *)bucket, (u32)(ops.xyzz_bytes / 4),
                           (u32 *)acc.buf[B_SCAN].p);
    } else {
        hipLaunchKernelGGL(k_msm_prep1, dim3((s.nkeys + 63) / 64), dim3(64), 0, st,   // single-wave workgroups, like the scans
                           s, S, L1, A.start, A.cnt, A.items, (u32 *)bucket, (u32)(ops.xyzz_bytes / 4));
    }
    MI_CHECK_HIP(ctx, hipGetLastError());
    if (acc.accum_gate) {   // the caller's condition for the heavy part (prove.hip: computeH first)
        const hipEvent_t g = (*acc.accum_gate)();
        acc.accum_gate = nullptr;
        if (g) MI_CHECK_HIP(ctx, hipStreamWaitEvent(st, g, 0));
    }
    MI_TRY(run_levels(ctx, ops, acc, s.nkeys, A, B, T_bound / L1 + s.nkeys + 1, max_count, L1, L2, pts, sorted, nullptr, bucket, timed, rprime, prep_sums));
    // everything below reads the bucket sums only: a point-sharded MSM (group.hip, SURVEY 8e option ii) stops here, exchanges
    // bucket slices between the devices and calls mi_msm_reduce_enqueue afterwards
    acc.tail_seg = seg;
    acc.entries_src = (const u32 *)((const char *)srt.host_wsum + 128 * 256);   // HOST word: the sort's key scan left the number of sorted entries there (fetch_max_and_scan_keys)
    acc.timed = timed;
    acc.deferred = defer_reduce;
    if (defer_reduce) {
        MI_CHECK_HIP(ctx, hipEventRecord(acc.ev[5], st));
        return MI_OK;
    }
    return msm_tail_enqueue(ctx, ops, acc);
}
// bucket reduce -> per-window partials -> window sums -> pinned host memory
static int32_t msm_tail_enqueue(mi_ctx *ctx, const MsmCurveOps &ops, MsmSlot &acc) {
    const MsmShape s = key_shape(acc);
    const u32 seg = acc.tail_seg;
    hipStream_t st = acc.stream;
    void *bucket = acc.buf[B_BUCKET].p;
    const u32 tb = (s.nbuckets + seg - 1) / seg;
    size_t win_pts = (size_t)s.nwin * tb + s.nwin + 1;
    for (u32 k = tb; k > 1; k = (k + ops.sum_T - 1) / ops.sum_T) win_pts += (size_t)s.nwin * ((k + ops.sum_T - 1) / ops.sum_T);
    MI_TRY(mi_reserve(ctx, acc.buf[B_WIN], win_pts * ops.xyzz_bytes));
    char *P = (char *)acc.buf[B_WIN].p;
    ops.bucket_reduce(st, (tb + 63) / 64, s.nwin, bucket, s.nbuckets, seg, tb, P);
    MI_CHECK_HIP(ctx, hipGetLastError());
    // window sums: LDS tree sums of sum_T partials per workgroup until one point per window is left ([w][0] layout = wsum[w])
    // (the LAST tree -- one point per window left -- stores straight into the slot's pinned host memory: no copy launch behind it)
    char *cur = P, *next = P + (size_t)s.nwin * tb * ops.xyzz_bytes;
    bool on_host = false;
    for (u32 k = tb; k > 1;) {
        const u32 nout = (k + ops.sum_T - 1) / ops.sum_T;
        ops.sum_tree(st, nout, s.nwin, cur, k, nout == 1 ? (char *)acc.host_wsum : next);
        on_host = nout == 1;
        cur = next; next += (size_t)s.nwin * nout * ops.xyzz_bytes; k = nout;
    }
    MI_CHECK_HIP(ctx, hipGetLastError());
    if (!on_host) MI_CHECK_HIP(ctx, hipMemcpyAsync(acc.host_wsum, P, ops.xyzz_bytes * s.nwin, hipMemcpyDeviceToHost, st));   // tb == 1: no tree ran
    MI_CHECK_HIP(ctx, hipEventRecord(acc.ev[4], st));
    acc.deferred = false;
    acc.active = true;
    return MI_OK;
}

static int32_t msm_finish(mi_ctx *ctx, const MsmCurveOps &ops, MsmSlot &sl, void *out) {
    if (sl.deferred) MI_FAIL(ctx, MI_EINVAL, "msm: finish before the deferred reduce was enqueued");
    if (!sl.active) { ops.combine_windows(nullptr, 0, 0, out); return MI_OK; }   // zero windows -> infinity
    MI_CHECK_HIP(ctx, hipStreamSynchronize(sl.stream));
    const MsmShape s = key_shape(sl);
    ops.combine_windows(sl.host_wsum, s.nwin, s.c, out);   // Horner on the host, <= 128 points
    if (sl.timed) {
        float ms = 0;
        MI_CHECK_HIP(ctx, hipEventElapsedTime(&ms, sl.ev[1], sl.ev[2]));
        ctx->stats.g1_accum_kernel_ms += ms;
        ctx->stats.g1_accum_pairs += sl.stat_pairs;
        ctx->stats.g1_accum_launches += 1;
        ctx->stats.g1_accum_entries += *sl.entries_src;
    }
    if (ops.xyzz_bytes == 128) ctx->stats.g1_level1_additions += *sl.entries_src;   // every G1 MSM, timed or not (a host word: the owning sort's key scan stored it)
    sl.active = false;
    return MI_OK;
}

// internal entry points used by prove.hip (curve: 1 = G1, 2 = G2)
int32_t mi_msm_precompute(mi_ctx *ctx, int curve, const void *base_dev, void *pre_dev, size_t n, uint32_t c) {
    if (!ctx || !base_dev || !pre_dev || c < 17 || c > 22 || n >= ((size_t)1 << 31)) return MI_EINVAL;
    const MsmCurveOps &ops = curve == 1 ? msm_g1_ops() : msm_g2_ops();
    if (!n) return MI_OK;
    // batched conversions (one inversion per 16 points per window instead of one per point: ~2.7x less work) need n XYZZ + n coordinates
    // of scratch for the duration of the build; without room for them the one-kernel form runs
    void *state = nullptr, *prefix = nullptr;
    const bool batched = !knobs_of(ctx)->precompute_unbatched && hipMalloc(&state, n * ops.xyzz_bytes) == hipSuccess && hipMalloc(&prefix, n * ops.coord_bytes) == hipSuccess;
    if (!batched) (void)hipGetLastError();
    if (batched) ops.precompute_batched(ctx->stream, base_dev, pre_dev, (u32)n, c, (256 + c - 1) / c, state, prefix);
    else ops.precompute(ctx->stream, base_dev, pre_dev, (u32)n, c, (256 + c - 1) / c);
    hipError_t e = hipGetLastError();
    if (batched && e == hipSuccess) e = hipStreamSynchronize(ctx->stream);   // the scratch goes away below
    if (state) (void)hipFree(state);
    if (prefix) (void)hipFree(prefix);
    MI_CHECK_HIP(ctx, e);
    return MI_OK;
}
int32_t mi_msm_enqueue(mi_ctx *ctx, int slot, int sort_slot, int curve, const void *pts_dev, const void *scalars_dev, size_t n,
                       uint32_t flags, hipEvent_t wait_ev, bool timed, uint32_t precomp_c, size_t stat_pairs, uint32_t generic_c) {
    if (slot < 0 || slot >= MI_MSM_SLOTS || sort_slot >= MI_MSM_SLOTS) return MI_EINVAL;
    static const char *const range_names[MI_MSM_SLOTS] = {"mi.msm.A.enqueue", "mi.msm.B1.enqueue", "mi.msm.B2.enqueue", "mi.msm.K.enqueue", "mi.msm.Z.enqueue", "mi.msm.PoK.enqueue"};
    const MiRange range(range_names[slot]);
    if (n > ((size_t)1 << 27)) MI_FAIL(ctx, MI_EINVAL, "msm: n > 2^27 pairs per device not supported (shard the points)");
    MsmSlot &sl = ctx->msm[slot];
    const std::function<hipEvent_t()> *gate_once = sl.accum_gate;   // valid for this call only, whatever path it takes
    sl.accum_gate = nullptr;
    sl.active = false;
    sl.deferred = false;
    sl.stat_pairs = stat_pairs ? stat_pairs : n;
    const bool defer = (flags & MI_MSM_DEFER_REDUCE) != 0, exact = (flags & MI_MSM_EXACT_SIZE) != 0, rprime = (flags & MI_MSM_PTS_RPRIME) != 0;
    flags &= ~(MI_MSM_DEFER_REDUCE | MI_MSM_EXACT_SIZE | MI_MSM_PTS_RPRIME);
    if (n == 0) {
        if (!defer) return MI_OK;
        // An EMPTY deferred MSM (a rank of a point-sharded MSM without pairs: group.hip) still leaves a bucket array of the plan's shape
        // -- all infinity -- so that the bucket exchange and the reduce treat this rank like every other.
        const MsmCurveOps &ops = curve == 1 ? msm_g1_ops() : msm_g2_ops();
        sl.n = 0; sl.G = 1;
        if (sort_slot >= 0) {   // the shape of the (equally empty) sort it would have shared
            const MsmSlot &srt = ctx->msm[sort_slot];
            if (srt.n != 0) MI_FAIL(ctx, MI_EINVAL, "msm: shared sort has a different length");
            sl.c = srt.c; sl.nwin_digits = srt.nwin_digits; sl.nwin_keys = srt.nwin_keys;
        } else {
            sl.c = precomp_c ? precomp_c : (knobs_of(ctx)->c ? knobs_of(ctx)->c : (generic_c >= 2 && generic_c <= 16 ? generic_c : auto_c(1)));
            sl.nwin_digits = (256 + sl.c - 1) / sl.c;
            sl.nwin_keys = precomp_c ? 1 : sl.nwin_digits;
        }
        sl.entries_cap = 0;
        const MsmShape s = key_shape(sl);
        MI_TRY(mi_reserve(ctx, sl.buf[B_BUCKET], (size_t)s.nkeys * ops.xyzz_bytes + 64));
        if (wait_ev) MI_CHECK_HIP(ctx, hipStreamWaitEvent(sl.stream, wait_ev, 0));
        MI_CHECK_HIP(ctx, hipEventRecord(sl.ev[3], sl.stream));
        MI_CHECK_HIP(ctx, hipMemsetAsync(sl.buf[B_BUCKET].p, 0, (size_t)s.nkeys * ops.xyzz_bytes + 64, sl.stream));
        sl.tail_seg = knobs_of(ctx)->seg ? knobs_of(ctx)->seg : msm_auto_seg(s.nbuckets);
        static const u32 no_entries = 0;
        sl.entries_src = &no_entries;   // (a host word, like every slot's)
        sl.timed = false;
        sl.deferred = true;
        MI_CHECK_HIP(ctx, hipEventRecord(sl.ev[5], sl.stream));
        return MI_OK;
    }
    if (wait_ev) MI_CHECK_HIP(ctx, hipStreamWaitEvent(sl.stream, wait_ev, 0));
    MI_CHECK_HIP(ctx, hipEventRecord(sl.ev[3], sl.stream));
    MsmSlot &srt = sort_slot >= 0 ? ctx->msm[sort_slot] : sl;
    if (sort_slot < 0 && precomp_c) MI_TRY(msm2_sort_enqueue(ctx, sl, (const Fr *)scalars_dev, (u32)n, flags, precomp_c, false, exact));
    else if (sort_slot < 0) MI_TRY(msm_sort_enqueue(ctx, sl, (const Fr *)scalars_dev, (u32)n, flags, generic_c, exact));
    else if (srt.n != n) MI_FAIL(ctx, MI_EINVAL, "msm: shared sort has a different length");
    sl.accum_gate = gate_once;
    return msm_accum_enqueue(ctx, curve == 1 ? msm_g1_ops() : msm_g2_ops(), srt, sl, pts_dev, timed, defer, rprime);
}
bool mi_msm_limb29_enabled(mi_ctx *ctx) { return !knobs_of(ctx)->no_rprime; }
uint32_t mi_msm_auto_c(size_t n) { return auto_c(n ? (u32)n : 1u); }
const MsmCurveOps &mi_msm_ops(int curve) { return curve == 1 ? msm_g1_ops() : msm_g2_ops(); }
int32_t mi_msm_reduce_enqueue(mi_ctx *ctx, int slot, int curve) {
    if (slot < 0 || slot >= MI_MSM_SLOTS) return MI_EINVAL;
    MsmSlot &sl = ctx->msm[slot];
    if (!sl.deferred) return MI_OK;   // an empty MSM (n == 0) never got as far as its buckets
    return msm_tail_enqueue(ctx, curve == 1 ? msm_g1_ops() : msm_g2_ops(), sl);
}
int32_t mi_msm_bucket_view(mi_ctx *ctx, int slot, int curve, MsmBucketView *v) {
    if (slot < 0 || slot >= MI_MSM_SLOTS || !v) return MI_EINVAL;
    MsmSlot &sl = ctx->msm[slot];
    const MsmCurveOps &ops = curve == 1 ? msm_g1_ops() : msm_g2_ops();
    *v = MsmBucketView{};
    if (!sl.deferred) return MI_OK;
    const MsmShape s = key_shape(sl);
    v->bucket = sl.buf[B_BUCKET].p; v->nkeys = s.nkeys; v->xyzz_bytes = ops.xyzz_bytes; v->seg = sl.tail_seg;
    v->stream = sl.stream; v->ready = sl.ev[5]; v->c = s.c; v->nwin = s.nwin;
    return MI_OK;
}
int32_t mi_msm_finish(mi_ctx *ctx, int slot, int curve, void *out_xyzz_host) {
    if (slot < 0 || slot >= MI_MSM_SLOTS) return MI_EINVAL;
    static const char *const range_names[MI_MSM_SLOTS] = {"mi.msm.A.collect", "mi.msm.B1.collect", "mi.msm.B2.collect", "mi.msm.K.collect", "mi.msm.Z.collect", "mi.msm.PoK.collect"};
    const MiRange range(range_names[slot]);
    return msm_finish(ctx, curve == 1 ? msm_g1_ops() : msm_g2_ops(), ctx->msm[slot], out_xyzz_host);
}

template <class F, class JacT>
static void xyzz_to_jac_out(const XYZZ<F> &r, JacT *out) {
    Jac<F> j;
    if (r.is_inf()) j = Jac<F>{F::one(), F::one(), F::zero()};
    else { Affine<F> a = xyzz_to_affine(r); j = Jac<F>{a.x, a.y, F::one()}; }
    std::memcpy(out, &j, sizeof(j));
}
// one MSM through slot 0, ordered after everything already queued on ctx->stream
template <class F, class JacT>
static int32_t msm_dev_entry(mi_ctx *ctx, int curve, const void *pts_dev, const void *scalars_dev, size_t n, uint32_t flags, JacT *out) {
    std::memset(&ctx->stats, 0, sizeof(ctx->stats));
    // G1 with enough pairs to repay one pass over the bases: copy them into the R' packed form the 9 x 29-bit level-1 kernel
    // reads (64 B in, 64 B out per point: 0.2 ms per 2^23 points against ~15 ms of MSM)
    uint32_t rp = 0;
    if (n >= ((size_t)1 << (curve == 1 ? 16 : 14)) && !knobs_of(ctx)->no_rprime) {
        const MsmCurveOps &o = curve == 1 ? msm_g1_ops() : msm_g2_ops();
        MI_TRY(mi_reserve(ctx, ctx->ws[23], n * (curve == 1 ? 64 : 128) + 64));
        o.to_rprime(ctx->stream, ctx->ws[23].p, pts_dev, n);
        MI_CHECK_HIP(ctx, hipGetLastError());
        pts_dev = ctx->ws[23].p;
        rp = MI_MSM_PTS_RPRIME;
    }
    MI_CHECK_HIP(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
    MI_TRY(mi_msm_enqueue(ctx, 0, -1, curve, pts_dev, scalars_dev, n, flags | MI_MSM_EXACT_SIZE | rp, ctx->ev[0], curve == 1));
    if (n) MI_CHECK_HIP(ctx, hipEventRecord(ctx->ev[1], ctx->msm[0].stream));
    XYZZ<F> r;
    MI_TRY(mi_msm_finish(ctx, 0, curve, &r));
    if (n) MI_CHECK_HIP(ctx, hipEventElapsedTime(&ctx->stats.total_ms, ctx->ev[0], ctx->ev[1]));
    xyzz_to_jac_out<F>(r, out);
    return MI_OK;
}
template <class F, class AffT, class JacT>
static int32_t msm_host_entry(mi_ctx *ctx, int curve, const AffT *pts, const mi_fr *scalars, size_t n, uint32_t flags, JacT *out) {
    if (!ctx || !out || ((!pts || !scalars) && n) || (flags & ~1u)) return MI_EINVAL;
    MI_TRY(mi_reserve(ctx, ctx->ws[2], n * sizeof(AffT) + 64));
    MI_TRY(mi_reserve(ctx, ctx->ws[3], n * sizeof(mi_fr) + 64));
    if (n) {
        MI_CHECK_HIP(ctx, hipMemcpyAsync(ctx->ws[2].p, pts, n * sizeof(AffT), hipMemcpyHostToDevice, ctx->stream));
        MI_CHECK_HIP(ctx, hipMemcpyAsync(ctx->ws[3].p, scalars, n * sizeof(mi_fr), hipMemcpyHostToDevice, ctx->stream));
    }
    return msm_dev_entry<F>(ctx, curve, ctx->ws[2].p, ctx->ws[3].p, n, flags, out);
}

template <class F, class JacT>
static int32_t msm_fixed_dev_entry(mi_ctx *ctx, int curve, const void *pre_dev, const void *scalars_dev, size_t n, uint32_t c, uint32_t flags, JacT *out) {
    std::memset(&ctx->stats, 0, sizeof(ctx->stats));
    const uint32_t rp = (flags & MI_MSM_TABLE_RPRIME) ? MI_MSM_PTS_RPRIME : 0;   // the table was converted by mi_msm_table_to_rprime_*: level 1 in 29-bit limbs
    flags &= ~MI_MSM_TABLE_RPRIME;
    MI_CHECK_HIP(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
    MI_TRY(mi_msm_enqueue(ctx, 0, -1, curve, pre_dev, scalars_dev, n, flags | MI_MSM_EXACT_SIZE | rp, ctx->ev[0], curve == 1, c));
    if (n) MI_CHECK_HIP(ctx, hipEventRecord(ctx->ev[1], ctx->msm[0].stream));
    XYZZ<F> r;
    MI_TRY(mi_msm_finish(ctx, 0, curve, &r));
    if (n) MI_CHECK_HIP(ctx, hipEventElapsedTime(&ctx->stats.total_ms, ctx->ev[0], ctx->ev[1]));
    xyzz_to_jac_out<F>(r, out);
    return MI_OK;
}
static int32_t table_to_rprime(mi_ctx *ctx, int curve, void *pre_dev, size_t n_points) {
    if (!ctx || (!pre_dev && n_points)) return MI_EINVAL;
    if (knobs_of(ctx)->no_rprime) MI_FAIL(ctx, MI_EINVAL, "msm: the 29-bit level-1 kernels are switched off on this context (mi_debug_set_msm_limb29)");
    mi_msm_ops(curve).to_rprime(ctx->stream, pre_dev, pre_dev, n_points);
    MI_CHECK_HIP(ctx, hipGetLastError());
    MI_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MI_OK;
}

extern "C" {
int32_t mi_msm_precompute_g1_dev(mi_ctx *ctx, const mi_g1_affine *base_dev, size_t n, uint32_t c, mi_g1_affine *pre_dev) {
    MI_TRY(mi_msm_precompute(ctx, 1, base_dev, pre_dev, n, c));
    MI_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MI_OK;
}
int32_t mi_msm_precompute_g2_dev(mi_ctx *ctx, const mi_g2_affine *base_dev, size_t n, uint32_t c, mi_g2_affine *pre_dev) {
    MI_TRY(mi_msm_precompute(ctx, 2, base_dev, pre_dev, n, c));
    MI_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MI_OK;
}
int32_t mi_msm_table_to_rprime_g1_dev(mi_ctx *ctx, mi_g1_affine *pre_dev, size_t n_points) { return table_to_rprime(ctx, 1, pre_dev, n_points); }
int32_t mi_msm_table_to_rprime_g2_dev(mi_ctx *ctx, mi_g2_affine *pre_dev, size_t n_points) { return table_to_rprime(ctx, 2, pre_dev, n_points); }
int32_t mi_msm_g1_fixed_dev(mi_ctx *ctx, const mi_g1_affine *pre_dev, const mi_fr *scalars_dev, size_t n, uint32_t c, uint32_t flags, mi_g1_jac *out) {
    if (!ctx || !out || ((!pre_dev || !scalars_dev) && n) || (flags & ~3u)) return MI_EINVAL;
    return msm_fixed_dev_entry<Fp>(ctx, 1, pre_dev, scalars_dev, n, c, flags, out);
}
int32_t mi_msm_g2_fixed_dev(mi_ctx *ctx, const mi_g2_affine *pre_dev, const mi_fr *scalars_dev, size_t n, uint32_t c, uint32_t flags, mi_g2_jac *out) {
    if (!ctx || !out || ((!pre_dev || !scalars_dev) && n) || (flags & ~3u)) return MI_EINVAL;
    return msm_fixed_dev_entry<Fp2>(ctx, 2, pre_dev, scalars_dev, n, c, flags, out);
}
int32_t mi_debug_set_msm_limb29(mi_ctx *ctx, uint32_t on) {
    if (!ctx || on > 2) return MI_EINVAL;
    knobs_of(ctx)->no_rprime = on ? 0 : 1;
    knobs_of(ctx)->std_partials = on == 2 ? 1 : 0;
    return MI_OK;
}
int32_t mi_debug_set_msm_precompute_batched(mi_ctx *ctx, uint32_t on) {
    if (!ctx || on > 1) return MI_EINVAL;
    knobs_of(ctx)->precompute_unbatched = on ? 0 : 1;
    return MI_OK;
}
int32_t mi_debug_set_msm_batch_affine(mi_ctx *ctx, uint32_t rounds) {
    if (!ctx || rounds > 4) return MI_EINVAL;
    knobs_of(ctx)->ba_rounds = rounds;
    return MI_OK;
}
int32_t mi_debug_set_msm_l1_waves(mi_ctx *ctx, uint32_t waves) {
    if (!ctx || (waves != 2 && waves != 3)) return MI_EINVAL;
    knobs_of(ctx)->l1_waves = waves;
    return MI_OK;
}
int32_t mi_debug_set_knob(mi_ctx *ctx, const char *name, int64_t value) {
    if (!ctx || !name) return MI_EINVAL;
    MsmKnobs *k = knobs_of(ctx);
    const auto is = [name](const char *n) { return std::strcmp(name, n) == 0; };
    const bool wg_ok = value == 1 || value == 2 || value == 4;
    if (is("l1_wg") && wg_ok) k->l1_wg = (u32)value;
    else if (is("g2_wg") && wg_ok) k->g2_wg = (u32)value;
    else if (is("l1_waves") && (value == 2 || value == 3)) k->l1_waves = (u32)value;
    else if (is("z_waves") && (value == 0 || value == 2)) k->z_waves = (u32)value;
    else if (is("g1_grid_per_cu") && value >= 0 && value <= 65536) k->g1_grid_per_cu = (u32)value;
    else if (is("g2_grid_per_cu") && value >= 0 && value <= 65536) k->g2_grid_per_cu = (u32)value;
    else if (is("count_per") && value >= 0 && value <= 64) k->count_per = (u32)value;
    else if (is("plain_scatter") && (value == 0 || value == 1)) k->plain_scatter = (u32)value;
    else if (is("z_count_fused") && (value == 0 || value == 1)) k->z_count_fused = (u32)value;
    else if (is("flat_item_l1") && (value == 0 || value == 1 || (value >= 4 && value <= 64))) k->flat_L1 = (u32)value;
    else if (is("dense_item_l1") && (value == 0 || value == 1 || (value >= 4 && value <= 64))) k->dense_L1 = (u32)value;
    else if (is("finisher") && (value == 0 || value == 1)) k->finisher = (u32)value;
    else if (is("finisher_max") && value >= 0 && value <= (1 << 20)) k->finisher_max = (u32)value;
    else if (is("item_l1") && (value == 0 || (value >= 2 && value <= 64))) k->L1 = (u32)value;     // entries per level-1 item (0 = 16)
    else if (is("item_l2") && (value == 0 || (value >= 2 && value <= 64))) k->L2 = (u32)value;     // partial sums per item of the later levels (0 = 8)
    else if (is("reduce_seg") && value >= 0 && value <= 256) k->seg = (u32)value;                    // buckets per bucket-reduce thread (0 = 8)
    else if (is("hold_accum") && (value == 0 || value == 1)) ctx->hold_accum = (uint32_t)value;   // = mi_debug_set_prove_schedule
    else if (is("finisher_min_level") && value >= 0 && value <= 16) k->finisher_min_level = (u32)value;
    else if (mi_ntt_set_knob(ctx, name, value)) return MI_OK;
    else MI_FAIL(ctx, MI_EINVAL, std::string("mi_debug_set_knob: unknown knob or value out of range: ") + name);
    return MI_OK;
}
int32_t mi_debug_set_msm_bound_levels(mi_ctx *ctx, uint32_t on) {
    if (!ctx || on > 1) return MI_EINVAL;
    knobs_of(ctx)->bound_levels = on;
    return MI_OK;
}
int32_t mi_debug_set_msm_one_pass_sort(mi_ctx *ctx, uint32_t on) {
    if (!ctx || on > 1) return MI_EINVAL;
    knobs_of(ctx)->one_pass_sort = on;
    return MI_OK;
}
int32_t mi_debug_set_msm_chunk(mi_ctx *ctx, uint32_t chunk) {
    if (!ctx) return MI_EINVAL;
    knobs_of(ctx)->chunk = chunk;
    return MI_OK;
}
int32_t mi_debug_set_msm_group_bits(mi_ctx *ctx, uint32_t gbits) {
    if (!ctx || (gbits && (gbits < 6 || gbits > 15))) return MI_EINVAL;
    knobs_of(ctx)->gbits = gbits;
    return MI_OK;
}
int32_t mi_debug_set_msm_plan(mi_ctx *ctx, uint32_t c, uint32_t L1, uint32_t L2, uint32_t seg, uint32_t G) {
    if (!ctx || c == 1 || c > 16 || G > 1024 || L1 == 1 || L2 == 1) return MI_EINVAL;   // items of one entry would never shrink a level
    MsmKnobs *k = knobs_of(ctx);
    k->c = c; k->L1 = L1; k->L2 = L2; k->seg = seg; k->G = G;
    return MI_OK;
}
int32_t mi_debug_set_prove_schedule(mi_ctx *ctx, uint32_t hold_accum) {
    if (!ctx || hold_accum > 1) return MI_EINVAL;
    ctx->hold_accum = hold_accum;
    return MI_OK;
}
int32_t mi_debug_set_prove_fixed_base(mi_ctx *ctx, uint32_t c_ak, uint32_t c_b, uint32_t c_z) {
    if (!ctx) return MI_EINVAL;
    for (uint32_t c : {c_ak, c_b, c_z}) if (c > 1 && (c < 17 || c > 22)) return MI_EINVAL;
    ctx->fixed_knob[0] = c_ak; ctx->fixed_knob[1] = c_b; ctx->fixed_knob[2] = c_z;
    return MI_OK;
}
int32_t mi_msm_g1(mi_ctx *ctx, const mi_g1_affine *pts, const mi_fr *scalars, size_t n, uint32_t flags, mi_g1_jac *out) {
    return msm_host_entry<Fp>(ctx, 1, pts, scalars, n, flags, out);
}
int32_t mi_msm_g2(mi_ctx *ctx, const mi_g2_affine *pts, const mi_fr *scalars, size_t n, uint32_t flags, mi_g2_jac *out) {
    return msm_host_entry<Fp2>(ctx, 2, pts, scalars, n, flags, out);
}
int32_t mi_msm_g1_dev(mi_ctx *ctx, const mi_g1_affine *pts_dev, const mi_fr *scalars_dev, size_t n, uint32_t flags, mi_g1_jac *out) {
    if (!ctx || !out || ((!pts_dev || !scalars_dev) && n) || (flags & ~1u)) return MI_EINVAL;
    return msm_dev_entry<Fp>(ctx, 1, pts_dev, scalars_dev, n, flags, out);
}
int32_t mi_msm_g2_dev(mi_ctx *ctx, const mi_g2_affine *pts_dev, const mi_fr *scalars_dev, size_t n, uint32_t flags, mi_g2_jac *out) {
    if (!ctx || !out || ((!pts_dev || !scalars_dev) && n) || (flags & ~1u)) return MI_EINVAL;
    return msm_dev_entry<Fp2>(ctx, 2, pts_dev, scalars_dev, n, flags, out);
}
}
