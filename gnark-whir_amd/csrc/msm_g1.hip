// G1 instantiation of the MSM point kernels (see msm_curve_kernels.cuh, msm.hip) + the level-1 accumulation in the
// 9 x 29-bit representation (curve29.cuh).
#include "msm_curve_kernels.cuh"
#include "curve29.cuh"
#include "msm_ba_g1.cuh"

// Level-1 bucket accumulation over points in the R' packed form: the item decomposition of k_msm_accum_affine, the mixed
// additions of an item in nine 29-bit limbs (162 multiplications and no carry instruction per product instead of 136 + 120;
// lazy additions), the item's sum converted back to the standard XYZZ once at its end.  +13..16 % mixed additions per second
// (tools/bench_limb29/madd29.hip: 13.1 against 11.3 G/s on L2-resident points, conversions included).
// Built twice.  Three waves per SIMD (168 VGPRs: 504 of a SIMD's 512 registers) is the default: alone on the GPU it is the faster
// build (14.0 against 13.5 G additions/s; N = 2^26, one proof filling the GPU: 3.86 against 3.76 proofs/s) and its launches are the
// short ones the roofline line is quoted on.  Two waves (196 VGPRs, no scratch; flag bit 1, mi_debug_set_msm_l1_waves) leaves room for
// a wave of the NTT passes (86 VGPRs) on the same SIMD, which fills the gaps the gathers leave: with three proofs in flight at
// N = 2^23 the JOB is 1.7 % faster and the single-proof latency 1 ms shorter (same-box A/B 32.9-33.0 against 32.0-32.6 proofs/s) while
// every launch of this kernel takes 8.8 instead of 5.5 ms.  Four waves would spill (332 B of scratch: 28.5 proofs/s).
static __device__ __forceinline__ void msm_accum_affine29_body(const G1Aff *pts, const u32 *sorted, const uint4 *tab, const u32 *item_start, u32 nkeys,
                                                               G1X *bucket, G1X *partial_out, u32 rp_partials) {
    const u32 total = item_start[nkeys], stride = gridDim.x * blockDim.x;
    for (u32 item = blockIdx.x * blockDim.x + threadIdx.x; item < total; item += stride) {
        const uint4 rec = tab[item];
        const u32 key = rec.x, b = rec.y, e = rec.z;
        G1X29 acc = g1x29_inf();
        // software pipelining: the gather of entry k + 1 is in flight while entry k is added (the kernel runs 2 or 3 waves per SIMD --
        // 196 / 168 VGPRs -- so the ~2 us of a random 64-B HBM read are not hidden by other waves alone)
        u32 v = sorted[b];
        const uint4 *q4 = reinterpret_cast<const uint4 *>(pts + (v & 0x7fffffffu));
        uint4 q0 = q4[0], q1 = q4[1], q2 = q4[2], q3 = q4[3];   // 64 B: x | y
        for (u32 k = b; k < e; k++) {
            const u32 w[16] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
            const bool neg = (v >> 31) != 0;
            if (k + 1 < e) {
                v = sorted[k + 1];
                q4 = reinterpret_cast<const uint4 *>(pts + (v & 0x7fffffffu));
                q0 = q4[0]; q1 = q4[1]; q2 = q4[2]; q3 = q4[3];
            }
            g1x29_madd(acc, w, neg);
        }
        // a bucket's only item leaves in the standard form (what the bucket reduce reads); a partial sum stays in the R' form for
        // the next level (k_msm_accum_xyzz29) -- a pack instead of four conversion products
        if (rec.w) bucket[key] = g1x29_to_std(acc);
        else if (rp_partials) g1x29_store_rp(acc, reinterpret_cast<u32 *>(partial_out + item));
        else partial_out[item] = g1x29_to_std(acc);
    }
}
// WG = waves per workgroup (1, 2, 4), WPS = waves per SIMD the registers are budgeted for (3: 168 VGPRs, 2: 196).  A workgroup of four
// waves takes one slot on each SIMD of a CU and gives all four back together: the multi-wave workgroups of the other streams (sorts,
// NTT passes, sum trees) then find a CU with room on every SIMD at once instead of waiting for the end of the launch.
template <int WG, int WPS>
__global__ void __launch_bounds__(64 * WG, WPS) k_msm_accum_affine29(const G1Aff *pts, const u32 *sorted, const uint4 *tab, const u32 *item_start, u32 nkeys,
                                                                    G1X *bucket, G1X *partial_out, u32 rp_partials) {
    msm_accum_affine29_body(pts, sorted, tab, item_start, nkeys, bucket, partial_out, rp_partials);
}
// Levels >= 2 of the item machinery over partial sums in the packed R' form: k_msm_accum_xyzz's decomposition, the additions in
// nine 29-bit limbs (g1x29_add), no conversion on the way in, one on the way out only for a bucket's final sum.
__global__ void __launch_bounds__(64, 3) k_msm_accum_xyzz29(const G1X *partial_in, const u32 *start, const u32 *cnt, const u32 *items, const u32 *item_start,
                                                         u32 nkeys, G1X *bucket, G1X *partial_out) {
    const u32 total = item_start[nkeys], stride = gridDim.x * blockDim.x;
    for (u32 item = blockIdx.x * blockDim.x + threadIdx.x; item < total; item += stride) {
        const u32 key = msm_item_key(item_start, nkeys, item);
        u32 b, e;
        msm_item_range(start[key], cnt[key], items[key], item - item_start[key], b, e);
        G1X29 acc = g1x29_load_rp(reinterpret_cast<const u32 *>(partial_in + b));
        for (u32 k = b + 1; k < e; k++) g1x29_add(acc, g1x29_load_rp(reinterpret_cast<const u32 *>(partial_in + k)));
        if (items[key] == 1) bucket[key] = g1x29_to_std(acc);
        else g1x29_store_rp(acc, reinterpret_cast<u32 *>(partial_out + item));
    }
}
static void launch_accum_xyzz29(hipStream_t st, unsigned grid, const void *pin, const u32 *start, const u32 *cnt, const u32 *items, const u32 *item_start,
                                u32 nkeys, u32 L, void *bucket, void *pout) {
    hipLaunchKernelGGL(k_msm_accum_xyzz29, dim3(grid), dim3(64), 0, st, (const G1X *)pin, start, cnt, items, item_start, nkeys, (G1X *)bucket, (G1X *)pout);
}
static void launch_accum_affine29(hipStream_t st, unsigned grid, const void *pts, const u32 *sorted, const u32 *start, const u32 *cnt, const u32 *items,
                                  const u32 *item_start, u32 nkeys, u32 L, void *bucket, void *pout, void *item_tab, u32 rp_partials, hipEvent_t ev_before) {
    hipLaunchKernelGGL(k_msm_item_table<Fp>, dim3(grid < 32768 ? grid : 32768), dim3(64), 0, st, start, cnt, items, item_start, nkeys, (uint4 *)item_tab);
    if (ev_before) (void)hipEventRecord(ev_before, st);
    // rp_partials: bit 0 = partial sums stay in the R' form, bit 1 = the two-waves-per-SIMD build, bits 2..3 = log2 of the waves per workgroup
    const u32 wg_log = (rp_partials >> 2) & 3u, wg = 1u << wg_log;
    const unsigned g = (grid + wg - 1) / wg;   // `grid` counts waves
#define MI_L1(WG, WPS) hipLaunchKernelGGL((k_msm_accum_affine29<WG, WPS>), dim3(g), dim3(64 * WG), 0, st, (const G1Aff *)pts, sorted, (const uint4 *)item_tab, \
                                          item_start, nkeys, (G1X *)bucket, (G1X *)pout, rp_partials & 1u)
    if (rp_partials & 2) { if (wg == 4) MI_L1(4, 2); else if (wg == 2) MI_L1(2, 2); else MI_L1(1, 2); }
    else { if (wg == 4) MI_L1(4, 3); else if (wg == 2) MI_L1(2, 3); else MI_L1(1, 3); }
#undef MI_L1
}
// one batch-affine round (msm_ba_g1.cuh)
template <int R>
static void ba_round(hipStream_t st, unsigned grid_cap, const G1Aff *pts, const u32 *sorted, const uint4 *tab, const u32 *item_start, u32 nkeys, uint64_t items_bound,
                     u32 target_waves, uint4 *nodes, uint4 *prefix, uint4 *totals, uint4 *invs) {
    const uint64_t slots = items_bound << (BA_LOG_L - R);
    const u32 K = msm_ba_K(slots, target_waves);
    const uint64_t chunks = msm_ba_chunks(slots, target_waves);
    const unsigned grid = (unsigned)(chunks < grid_cap ? (chunks ? chunks : 1) : grid_cap);
    hipLaunchKernelGGL(k_ba_fwd<R>, dim3(grid), dim3(64), 0, st, pts, sorted, tab, item_start, nkeys, (const uint4 *)nodes, prefix, totals, K);
    hipLaunchKernelGGL(k_ba_inv, dim3((unsigned)((chunks + 63) / 64)), dim3(64), 0, st, item_start, nkeys, BA_LOG_L - R, K, (const u32 *)totals, (u32 *)invs);
    hipLaunchKernelGGL(k_ba_bwd<R>, dim3(grid), dim3(64), 0, st, pts, sorted, tab, item_start, nkeys, nodes, (const uint4 *)prefix, (const uint4 *)invs, K);
}
static void launch_accum_affine_ba(hipStream_t st, unsigned grid_cap, const void *pts_, const u32 *sorted, const u32 *start, const u32 *cnt, const u32 *items,
                                   const u32 *item_start, u32 nkeys, void *bucket, void *pout, void *item_tab, u32 rp_partials, u32 rounds, uint64_t items_bound,
                                   u32 target_waves, void *nodes_, void *prefix_, void *totals_, void *invs_, hipEvent_t ev_before) {
    const G1Aff *pts = (const G1Aff *)pts_;
    const uint4 *tab = (const uint4 *)item_tab;
    uint4 *nodes = (uint4 *)nodes_, *prefix = (uint4 *)prefix_, *totals = (uint4 *)totals_, *invs = (uint4 *)invs_;
    unsigned tgrid = (unsigned)((items_bound + 255) / 256);
    hipLaunchKernelGGL(k_msm_item_table<Fp>, dim3(tgrid < 8192 ? (tgrid ? tgrid : 1) : 8192), dim3(256), 0, st, start, cnt, items, item_start, nkeys, (uint4 *)item_tab);
    if (ev_before) (void)hipEventRecord(ev_before, st);
    if (rounds >= 1) ba_round<1>(st, grid_cap, pts, sorted, tab, item_start, nkeys, items_bound, target_waves, nodes, prefix, totals, invs);
    if (rounds >= 2) ba_round<2>(st, grid_cap, pts, sorted, tab, item_start, nkeys, items_bound, target_waves, nodes, prefix, totals, invs);
    if (rounds >= 3) ba_round<3>(st, grid_cap, pts, sorted, tab, item_start, nkeys, items_bound, target_waves, nodes, prefix, totals, invs);
    if (rounds >= 4) ba_round<4>(st, grid_cap, pts, sorted, tab, item_start, nkeys, items_bound, target_waves, nodes, prefix, totals, invs);
    unsigned fgrid = (unsigned)((items_bound + 63) / 64);
    if (fgrid > grid_cap) fgrid = grid_cap;
    if (!fgrid) fgrid = 1;
    G1X *bk = (G1X *)bucket, *po = (G1X *)pout;
    const u32 rpp = rp_partials & 1u;
    switch (rounds) {
    case 1: hipLaunchKernelGGL(k_ba_finish<1>, dim3(fgrid), dim3(64), 0, st, pts, sorted, tab, item_start, nkeys, (const uint4 *)nodes, bk, po, rpp); break;
    case 2: hipLaunchKernelGGL(k_ba_finish<2>, dim3(fgrid), dim3(64), 0, st, pts, sorted, tab, item_start, nkeys, (const uint4 *)nodes, bk, po, rpp); break;
    case 3: hipLaunchKernelGGL(k_ba_finish<3>, dim3(fgrid), dim3(64), 0, st, pts, sorted, tab, item_start, nkeys, (const uint4 *)nodes, bk, po, rpp); break;
    default: hipLaunchKernelGGL(k_ba_finish<4>, dim3(fgrid), dim3(64), 0, st, pts, sorted, tab, item_start, nkeys, (const uint4 *)nodes, bk, po, rpp); break;
    }
}
// the finisher over partial sums in the packed R' form (k_msm_accum_xyzz29's additions)
struct FinG1rp {
    typedef G1X Partial;
    static constexpr int LDS_WORDS_PER_WAVE = 0;
    G1X29 acc;
    MI_D explicit FinG1rp(u32 *) {}
    MI_D void load(const Partial *p) { acc = g1x29_load_rp(reinterpret_cast<const u32 *>(p)); }
    MI_D void add(const Partial *p) { g1x29_add(acc, g1x29_load_rp(reinterpret_cast<const u32 *>(p))); }
    MI_D void store(Partial *p) const { g1x29_store_rp(acc, reinterpret_cast<u32 *>(p)); }
    MI_D void to_bucket(G1X *b) const { *b = g1x29_to_std(acc); }
};
static void launch_finish_g1(hipStream_t st, unsigned nb_small, unsigned nb_big, void *partials, const u32 *list_small, const u32 *list_big, const u32 *counters,
                             const u32 *item_start, const u32 *items, void *bucket, u32 rp) {
    // (three waves per SIMD = 168 VGPRs: a finisher workgroup fits where ONE four-wave level-1 workgroup has retired; with the compiler's
    //  free choice -- 208 -- it waited for a larger hole: 2.6 ms per launch inside the job, rocprofv3)
    if (rp) launch_finish_form<FinG1rp, G1X, 256, 3>(st, nb_small, nb_big, partials, list_small, list_big, counters, item_start, items, bucket);
    else launch_finish_form<FinStd<Fp>, G1X, 256, 3>(st, nb_small, nb_big, partials, list_small, list_big, counters, item_start, items, bucket);
}
__global__ void k_g1_to_rprime(G1Aff *dst, const G1Aff *src, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    G1Aff a = src[i];
    dst[i] = G1Aff{fe_to_rprime_packed(a.x), fe_to_rprime_packed(a.y)};   // (0, 0) = infinity stays (0, 0)
}
static void launch_to_rprime(hipStream_t st, void *dst, const void *src, size_t n) {
    if (n) hipLaunchKernelGGL(k_g1_to_rprime, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (G1Aff *)dst, (const G1Aff *)src, n);
}

const MsmCurveOps &msm_g1_ops() {
    static const MsmCurveOps ops = {sizeof(G1X), launch_accum_affine<Fp>, launch_accum_xyzz<Fp>, launch_bucket_reduce<Fp>, SumT<Fp>::value, launch_sum_tree<Fp>, launch_precompute<Fp>, launch_precompute_batched<Fp>, sizeof(Fp), host_combine_windows<Fp>, launch_sum_slices<Fp>, launch_accum_affine29, launch_accum_xyzz29, launch_accum_affine_ba, launch_finish_g1, 256, 4096, launch_to_rprime};
    return ops;
}
