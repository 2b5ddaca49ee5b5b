// Pippenger MSM on gfx950 for BN254 G1 and G2: kernels around the per-thread bodies of
// msm_core.cuh plus the (sync-free) host orchestration.  See msm_core.cuh for the algorithm and the
// reference functions it replaces (gnark-crypto MultiExp, reached from /root/reference/mt.go:496).
//
// HBM layout per MSM call (all grow-only ctx workspaces, n pairs, nwin windows, K = nwin*2^(c-1) keys):
//   digits  int16 [nwin][n]            window-major signed digits
//   H, S    u32   [K][G] (+1)          per-(key, slice) histogram and its flat exclusive scan
//   sorted  u32   [<= nwin*n]          point index | sign<<31, grouped by key
//   start/cnt/items/item_start u32 [K] item decomposition of the current level (ping-pong)
//   partial XYZZ  [items]              per-item partial sums (ping-pong between levels)
//   bucket  XYZZ  [K]                  final bucket sums (zero-filled = infinity)
// Roofline: the level-1 accumulate kernel reads 4 B + one 64 B (G1) / 128 B (G2) point per entry and
// performs one mixed addition (~10 Fp products): it is VALU-bound by two orders of magnitude; the HBM
// figure reported for it is the algorithmic 96 B (160 B) per pair of SURVEY 8d over its duration.
#include "ctx.h"
#include "msm_core.cuh"
#include <cstring>
#include <new>

struct MsmKnobs {
    u32 c = 0, L1 = 0, L2 = 0, seg = 0, G = 0;  // 0 = automatic
};
static MsmKnobs *knobs_of(mi_ctx *ctx) { return reinterpret_cast<MsmKnobs *>(ctx->msm_knobs); }
__global__ void k_msm_hist(MsmShape s, const int16_t *digits, u32 *H);
__global__ void k_msm_scatter(MsmShape s, const int16_t *digits, const u32 *S, u32 *sorted);
void mi_msm_state_init(mi_ctx *ctx) {
    new (ctx->msm_knobs) MsmKnobs();
    // the c = 16 histogram / cursor image is 128 KiB of LDS (gfx950 allows 160 KiB per workgroup)
    (void)hipFuncSetAttribute((const void *)k_msm_hist, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)k_msm_scatter, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

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
__global__ void __launch_bounds__(1024) k_msm_scatter(MsmShape s, const int16_t *digits, const u32 *S, u32 *sorted) {
    extern __shared__ u32 lds_u32[];
    const u32 g = blockIdx.x, w = blockIdx.y;
    msm_scatter_init(s, S, g, w, lds_u32, threadIdx.x, blockDim.x);
    __syncthreads();
    msm_scatter_move(s, digits, g, w, lds_u32, sorted, threadIdx.x, blockDim.x);
}
__global__ void k_msm_prep1(MsmShape s, const u32 *S, u32 L, u32 *start, u32 *cnt, u32 *items) {
    u32 key = blockIdx.x * blockDim.x + threadIdx.x;
    if (key < s.nkeys) msm_prep_level1(s, S, L, start, cnt, items, key);
}
__global__ void k_msm_prep_next(u32 nkeys, const u32 *prev_items, const u32 *prev_item_start, u32 L, u32 *start, u32 *cnt, u32 *items) {
    u32 key = blockIdx.x * blockDim.x + threadIdx.x;
    if (key < nkeys) msm_prep_next(prev_items, prev_item_start, L, start, cnt, items, key);
}
// keyed-by-window item decomposition of the bucket-reduce partials: window w owns [w*tb, (w+1)*tb)
__global__ void k_msm_prep_windows(u32 nwin, u32 tb, u32 L, u32 *start, u32 *cnt, u32 *items) {
    u32 w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w < nwin) { start[w] = w * tb; cnt[w] = tb; items[w] = (tb + L - 1) / L; }
}
template <class F>
__global__ void __launch_bounds__(64) k_msm_accum_affine(const Affine<F> *pts, const u32 *sorted, const u32 *start, const u32 *cnt,
                                                         const u32 *items, const u32 *item_start, u32 nkeys, u32 L,
                                                         XYZZ<F> *bucket, XYZZ<F> *partial_out) {
    u32 item = blockIdx.x * blockDim.x + threadIdx.x;
    msm_accum_affine_body<F>(pts, sorted, start, cnt, items, item_start, nkeys, L, bucket, partial_out, item);
}
template <class F>
__global__ void __launch_bounds__(64) k_msm_accum_xyzz(const XYZZ<F> *partial_in, const u32 *start, const u32 *cnt, const u32 *items,
                                                       const u32 *item_start, u32 nkeys, u32 L, XYZZ<F> *bucket, XYZZ<F> *partial_out) {
    u32 item = blockIdx.x * blockDim.x + threadIdx.x;
    msm_accum_xyzz_body<F>(partial_in, start, cnt, items, item_start, nkeys, L, bucket, partial_out, item);
}
template <class F>
__global__ void __launch_bounds__(64) k_msm_bucket_reduce(const XYZZ<F> *bucket, u32 nbuckets, u32 seg, u32 tb, XYZZ<F> *out) {
    u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < tb) msm_bucket_reduce_body<F>(bucket, nbuckets, seg, out, blockIdx.y, t);
}

// ---------------------------------------------------------------- exclusive scan of u32 (out has m+1 entries, out[m] = total)
static constexpr u32 SCAN_PER_THREAD = 16, SCAN_THREADS = 256, SCAN_BLOCK = SCAN_PER_THREAD * SCAN_THREADS;
__device__ u32 block_exclusive_scan_256(u32 v, u32 *lds, u32 *total) {
    // Hillis-Steele over 256 thread sums
    u32 tid = threadIdx.x;
    lds[tid] = v;
    __syncthreads();
    for (u32 off = 1; off < 256; off <<= 1) {
        u32 a = tid >= off ? lds[tid - off] : 0;
        __syncthreads();
        lds[tid] += a;
        __syncthreads();
    }
    u32 incl = lds[tid];
    *total = lds[255];
    __syncthreads();
    return incl - v;
}
__global__ void __launch_bounds__(256) k_scan_block_sums(const u32 *in, size_t m, u32 *block_sums) {
    __shared__ u32 lds[256];
    size_t base = (size_t)blockIdx.x * SCAN_BLOCK + (size_t)threadIdx.x * SCAN_PER_THREAD;
    u32 s = 0;
    for (u32 k = 0; k < SCAN_PER_THREAD; k++) if (base + k < m) s += in[base + k];
    u32 total;
    block_exclusive_scan_256(s, lds, &total);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}
__global__ void __launch_bounds__(256) k_scan_of_sums(u32 *block_sums, u32 nblocks) {  // single workgroup, in place
    __shared__ u32 lds[256];
    u32 carry = 0;
    for (u32 base = 0; base < nblocks; base += 256) {
        u32 i = base + threadIdx.x;
        u32 v = i < nblocks ? block_sums[i] : 0, total;
        u32 ex = block_exclusive_scan_256(v, lds, &total);
        if (i < nblocks) block_sums[i] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) block_sums[nblocks] = carry;
}
__global__ void __launch_bounds__(256) k_scan_final(const u32 *in, size_t m, const u32 *block_sums, u32 *out) {
    __shared__ u32 lds[256];
    size_t base = (size_t)blockIdx.x * SCAN_BLOCK + (size_t)threadIdx.x * SCAN_PER_THREAD;
    u32 v[SCAN_PER_THREAD], s = 0;
    for (u32 k = 0; k < SCAN_PER_THREAD; k++) { v[k] = base + k < m ? in[base + k] : 0; s += v[k]; }
    u32 total;
    u32 ex = block_exclusive_scan_256(s, lds, &total) + block_sums[blockIdx.x];
    for (u32 k = 0; k < SCAN_PER_THREAD; k++) {
        if (base + k < m) out[base + k] = ex;
        ex += v[k];
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) out[m] = block_sums[gridDim.x];
}
static int32_t exclusive_scan(mi_ctx *ctx, const u32 *in, size_t m, u32 *out, DevBuf &tmp) {
    u32 nblocks = (u32)((m + SCAN_BLOCK - 1) / SCAN_BLOCK);
    if (nblocks == 0) nblocks = 1;
    MI_TRY(mi_reserve(ctx, tmp, (size_t)(nblocks + 1) * 4));
    u32 *bs = (u32 *)tmp.p;
    hipLaunchKernelGGL(k_scan_block_sums, dim3(nblocks), dim3(256), 0, ctx->stream, in, m, bs);
    hipLaunchKernelGGL(k_scan_of_sums, dim3(1), dim3(256), 0, ctx->stream, bs, nblocks);
    hipLaunchKernelGGL(k_scan_final, dim3(nblocks), dim3(256), 0, ctx->stream, in, m, bs, out);
    MI_CHECK_HIP(ctx, hipGetLastError());
    return MI_OK;
}

// ---------------------------------------------------------------- orchestration
// workspace slots (ctx->ws): 4 digits, 5 H, 6 S, 7 sorted, 8 level arrays, 9/10 partial ping-pong,
// 11 buckets, 12 scan temp, 13 window partials / sums
struct LevelArrays { u32 *start, *cnt, *items, *item_start; };

static u32 auto_c(u32 n) {
    u32 lg = 0;
    while ((2u << lg) <= n) lg++;          // floor(log2 n) for n >= 1
    int c = (int)lg - 4;
    if (c < 3) c = 3;
    if (c > 16) c = 16;
    return (u32)c;
}

// Runs levels of the item machinery over `nkeys` keys whose level-0 decomposition (start/cnt/items) is
// already in cur.  first_affine: level 0 reads (pts, sorted); else level 0 reads partial_first.
template <class F>
static int32_t run_levels(mi_ctx *ctx, u32 nkeys, LevelArrays cur, LevelArrays nxt, u64 first_items_bound, u64 max_count, u32 L_first, u32 L_next,
                          const Affine<F> *pts, const u32 *sorted, const XYZZ<F> *partial_first, XYZZ<F> *final_out, bool time_first) {
    const XYZZ<F> *pin = partial_first;
    u64 items_bound = first_items_bound;
    u64 m = max_count;  // bound on entries of the largest key at this level
    u32 L = L_first;
    for (u32 level = 0;; level++) {
        MI_TRY(exclusive_scan(ctx, cur.items, nkeys, cur.item_start, ctx->ws[12]));
        DevBuf &pout_buf = ctx->ws[(level & 1) ? 10 : 9];
        MI_TRY(mi_reserve(ctx, pout_buf, (items_bound + 1) * sizeof(XYZZ<F>)));
        XYZZ<F> *pout = (XYZZ<F> *)pout_buf.p;
        u32 grid = (u32)((items_bound + 63) / 64);
        if (grid == 0) grid = 1;
        if (time_first && level == 0) MI_CHECK_HIP(ctx, hipEventRecord(ctx->ev[20], ctx->stream));
        if (level == 0 && pts)
            hipLaunchKernelGGL(k_msm_accum_affine<F>, dim3(grid), dim3(64), 0, ctx->stream, pts, sorted, cur.start, cur.cnt, cur.items, cur.item_start, nkeys, L, final_out, pout);
        else
            hipLaunchKernelGGL(k_msm_accum_xyzz<F>, dim3(grid), dim3(64), 0, ctx->stream, pin, cur.start, cur.cnt, cur.items, cur.item_start, nkeys, L, final_out, pout);
        MI_CHECK_HIP(ctx, hipGetLastError());
        if (time_first && level == 0) MI_CHECK_HIP(ctx, hipEventRecord(ctx->ev[21], ctx->stream));
        u64 m_next = (m + L - 1) / L;  // entries of the largest key at the next level
        if (m_next <= 1) break;
        hipLaunchKernelGGL(k_msm_prep_next, dim3((nkeys + 255) / 256), dim3(256), 0, ctx->stream, nkeys, cur.items, cur.item_start, L_next, nxt.start, nxt.cnt, nxt.items);
        MI_CHECK_HIP(ctx, hipGetLastError());
        // items at the next level: every continuing key has >= 2 entries, so items <= entries/L + keys
        u64 nb = items_bound / L_next + (items_bound < nkeys ? items_bound : nkeys) + 1;
        items_bound = nb < items_bound ? nb : items_bound;
        m = m_next; L = L_next; pin = pout;
        LevelArrays t = cur; cur = nxt; nxt = t;
    }
    return MI_OK;
}

template <class F>
static int32_t msm_run(mi_ctx *ctx, const Affine<F> *pts, const Fr *scalars, size_t n_sz, u32 flags, XYZZ<F> *result_host, bool count_stats) {
    if (n_sz == 0) { *result_host = XYZZ<F>::inf(); return MI_OK; }
    if (n_sz > ((size_t)1 << 27)) MI_FAIL(ctx, MI_EINVAL, "msm: n > 2^27 pairs per device not supported (shard the points)");
    const u32 n = (u32)n_sz;
    MsmKnobs *kn = knobs_of(ctx);
    const u32 c = kn->c ? kn->c : auto_c(n);
    u32 G = kn->G ? kn->G : (n / 8192 > 64 ? 64 : (n / 8192 ? n / 8192 : 1));
    const u32 L1 = kn->L1 ? kn->L1 : 32, L2 = kn->L2 ? kn->L2 : 16;
    const MsmShape s = msm_shape(n, c, G);
    const u32 seg = kn->seg ? kn->seg : (s.nbuckets >= 256 ? 8 : 2);
    const u64 T_bound = (u64)s.nwin * n;

    MI_TRY(mi_reserve(ctx, ctx->ws[4], T_bound * 2 + 64));
    MI_TRY(mi_reserve(ctx, ctx->ws[5], ((size_t)s.nkeys * G + 1) * 4));
    MI_TRY(mi_reserve(ctx, ctx->ws[6], ((size_t)s.nkeys * G + 1) * 4));
    MI_TRY(mi_reserve(ctx, ctx->ws[7], (T_bound + 1) * 4));
    MI_TRY(mi_reserve(ctx, ctx->ws[8], ((size_t)s.nkeys + 1) * 4 * 8));
    MI_TRY(mi_reserve(ctx, ctx->ws[11], (size_t)s.nkeys * sizeof(XYZZ<F>)));
    int16_t *digits = (int16_t *)ctx->ws[4].p;
    u32 *H = (u32 *)ctx->ws[5].p, *S = (u32 *)ctx->ws[6].p, *sorted = (u32 *)ctx->ws[7].p;
    u32 *la = (u32 *)ctx->ws[8].p;
    const size_t stride = (size_t)s.nkeys + 1;
    LevelArrays A{la, la + stride, la + 2 * stride, la + 3 * stride}, B{la + 4 * stride, la + 5 * stride, la + 6 * stride, la + 7 * stride};
    XYZZ<F> *bucket = (XYZZ<F> *)ctx->ws[11].p;

    // 1. digits
    hipLaunchKernelGGL(k_msm_digits, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, s, scalars, (flags & MI_MSM_SCALARS_CANONICAL) ? 0 : 1, digits);
    // 2. counting sort
    const size_t lds_bytes = (size_t)s.nbuckets * 4;
    hipLaunchKernelGGL(k_msm_hist, dim3(G, s.nwin), dim3(1024), lds_bytes, ctx->stream, s, digits, H);
    MI_CHECK_HIP(ctx, hipGetLastError());
    MI_TRY(exclusive_scan(ctx, H, (size_t)s.nkeys * G, S, ctx->ws[12]));
    hipLaunchKernelGGL(k_msm_scatter, dim3(G, s.nwin), dim3(1024), lds_bytes, ctx->stream, s, digits, S, sorted);
    // 3. accumulate
    MI_CHECK_HIP(ctx, hipMemsetAsync(bucket, 0, (size_t)s.nkeys * sizeof(XYZZ<F>), ctx->stream));
    hipLaunchKernelGGL(k_msm_prep1, dim3((s.nkeys + 255) / 256), dim3(256), 0, ctx->stream, s, S, L1, A.start, A.cnt, A.items);
    MI_CHECK_HIP(ctx, hipGetLastError());
    MI_TRY(run_levels<F>(ctx, s.nkeys, A, B, T_bound / L1 + s.nkeys + 1, n, L1, L2, pts, sorted, nullptr, bucket, count_stats));
    // 4. bucket reduce -> per-window partials -> window sums
    const u32 tb = (s.nbuckets + seg - 1) / seg;
    MI_TRY(mi_reserve(ctx, ctx->ws[13], ((size_t)s.nwin * tb + s.nwin + 1) * sizeof(XYZZ<F>)));
    XYZZ<F> *P = (XYZZ<F> *)ctx->ws[13].p, *wsum = P + (size_t)s.nwin * tb;
    hipLaunchKernelGGL(k_msm_bucket_reduce<F>, dim3((tb + 63) / 64, s.nwin), dim3(64), 0, ctx->stream, bucket, s.nbuckets, seg, tb, P);
    hipLaunchKernelGGL(k_msm_prep_windows, dim3(1), dim3(128), 0, ctx->stream, s.nwin, tb, L2, A.start, A.cnt, A.items);
    MI_CHECK_HIP(ctx, hipGetLastError());
    if (s.nwin > 128) MI_FAIL(ctx, MI_EINVAL, "msm: too many windows");
    MI_TRY(run_levels<F>(ctx, s.nwin, A, B, (u64)s.nwin * ((tb + L2 - 1) / L2) + 1, tb, L2, L2, (const Affine<F> *)nullptr, nullptr, P, wsum, false));
    // 5. window sums -> host, Horner
    XYZZ<F> hw[128];
    MI_CHECK_HIP(ctx, hipMemcpyAsync(hw, wsum, sizeof(XYZZ<F>) * s.nwin, hipMemcpyDeviceToHost, ctx->stream));
    MI_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *result_host = msm_combine_windows<F>(hw, s.nwin, s.c);
    if (count_stats) {
        float ms = 0;
        MI_CHECK_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev[20], ctx->ev[21]));
        ctx->stats.g1_accum_kernel_ms += ms;
        ctx->stats.g1_accum_pairs += n;
        ctx->stats.g1_accum_launches += 1;
    }
    return MI_OK;
}

int32_t mi_msm_g1_xyzz(mi_ctx *ctx, const void *pts_dev, const void *scalars_dev, size_t n, uint32_t flags, void *out_xyzz_host) {
    return msm_run<Fp>(ctx, (const G1Aff *)pts_dev, (const Fr *)scalars_dev, n, flags, (G1X *)out_xyzz_host, true);
}
int32_t mi_msm_g2_xyzz(mi_ctx *ctx, const void *pts_dev, const void *scalars_dev, size_t n, uint32_t flags, void *out_xyzz_host) {
    return msm_run<Fp2>(ctx, (const G2Aff *)pts_dev, (const Fr *)scalars_dev, n, flags, (G2X *)out_xyzz_host, false);
}

template <class F, class JacT>
static void xyzz_to_jac_out(const XYZZ<F> &r, JacT *out) {
    Jac<F> j;
    if (r.is_inf()) j = Jac<F>{F::one(), F::one(), F::zero()};
    else { Affine<F> a = xyzz_to_affine(r); j = Jac<F>{a.x, a.y, F::one()}; }
    std::memcpy(out, &j, sizeof(j));
}
template <class F, class AffT, class JacT>
static int32_t msm_host_entry(mi_ctx *ctx, const AffT *pts, const mi_fr *scalars, size_t n, uint32_t flags, JacT *out, bool g1) {
    if (!ctx || !out || ((!pts || !scalars) && n) || (flags & ~1u)) return MI_EINVAL;
    MI_TRY(mi_reserve(ctx, ctx->ws[2], n * sizeof(AffT) + 64));
    MI_TRY(mi_reserve(ctx, ctx->ws[3], n * sizeof(mi_fr) + 64));
    if (n) {
        MI_CHECK_HIP(ctx, hipMemcpyAsync(ctx->ws[2].p, pts, n * sizeof(AffT), hipMemcpyHostToDevice, ctx->stream));
        MI_CHECK_HIP(ctx, hipMemcpyAsync(ctx->ws[3].p, scalars, n * sizeof(mi_fr), hipMemcpyHostToDevice, ctx->stream));
    }
    std::memset(&ctx->stats, 0, sizeof(ctx->stats));
    XYZZ<F> r;
    MI_TRY(msm_run<F>(ctx, (const Affine<F> *)ctx->ws[2].p, (const Fr *)ctx->ws[3].p, n, flags, &r, g1));
    xyzz_to_jac_out<F>(r, out);
    return MI_OK;
}

extern "C" {
int32_t mi_debug_set_msm_plan(mi_ctx *ctx, uint32_t c, uint32_t L1, uint32_t L2, uint32_t seg, uint32_t G) {
    if (!ctx || c == 1 || c > 16 || G > 1024) return MI_EINVAL;
    MsmKnobs *k = knobs_of(ctx);
    k->c = c; k->L1 = L1; k->L2 = L2; k->seg = seg; k->G = G;
    return MI_OK;
}
int32_t mi_msm_g1(mi_ctx *ctx, const mi_g1_affine *pts, const mi_fr *scalars, size_t n, uint32_t flags, mi_g1_jac *out) {
    return msm_host_entry<Fp>(ctx, pts, scalars, n, flags, out, true);
}
int32_t mi_msm_g2(mi_ctx *ctx, const mi_g2_affine *pts, const mi_fr *scalars, size_t n, uint32_t flags, mi_g2_jac *out) {
    return msm_host_entry<Fp2>(ctx, pts, scalars, n, flags, out, false);
}
int32_t mi_msm_g1_dev(mi_ctx *ctx, const mi_g1_affine *pts_dev, const mi_fr *scalars_dev, size_t n, uint32_t flags, mi_g1_jac *out) {
    if (!ctx || !out || ((!pts_dev || !scalars_dev) && n) || (flags & ~1u)) return MI_EINVAL;
    std::memset(&ctx->stats, 0, sizeof(ctx->stats));
    MI_CHECK_HIP(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
    G1X r;
    MI_TRY(msm_run<Fp>(ctx, (const G1Aff *)pts_dev, (const Fr *)scalars_dev, n, flags, &r, true));
    MI_CHECK_HIP(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
    MI_CHECK_HIP(ctx, hipEventSynchronize(ctx->ev[1]));
    MI_CHECK_HIP(ctx, hipEventElapsedTime(&ctx->stats.total_ms, ctx->ev[0], ctx->ev[1]));
    xyzz_to_jac_out<Fp>(r, out);
    return MI_OK;
}
int32_t mi_msm_g2_dev(mi_ctx *ctx, const mi_g2_affine *pts_dev, const mi_fr *scalars_dev, size_t n, uint32_t flags, mi_g2_jac *out) {
    if (!ctx || !out || ((!pts_dev || !scalars_dev) && n) || (flags & ~1u)) return MI_EINVAL;
    std::memset(&ctx->stats, 0, sizeof(ctx->stats));
    MI_CHECK_HIP(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
    G2X r;
    MI_TRY(msm_run<Fp2>(ctx, (const G2Aff *)pts_dev, (const Fr *)scalars_dev, n, flags, &r, false));
    MI_CHECK_HIP(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
    MI_CHECK_HIP(ctx, hipEventSynchronize(ctx->ev[1]));
    MI_CHECK_HIP(ctx, hipEventElapsedTime(&ctx->stats.total_ms, ctx->ev[0], ctx->ev[1]));
    xyzz_to_jac_out<Fp2>(r, out);
    return MI_OK;
}
}
