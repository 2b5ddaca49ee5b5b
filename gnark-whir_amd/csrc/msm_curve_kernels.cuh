// Point kernels of the MSM, templated on the coordinate field; included by msm_g1.hip (F = Fp) and msm_g2.hip (F = Fp2).
#pragma once
#include <hip/hip_runtime.h>
#include "msm2_core.cuh"
#include "msm_curve_ops.h"
#include "batch_affine.cuh"

// Occupancy target per curve (measured with tools/bench_g2 on MI355X): the G1 mixed add needs ~60 VGPRs and runs
// 8 waves/SIMD; the G2 one wants > 256 -- 2 waves/SIMD with the Fp multiplier out of line (480 B of scratch) is the
// fastest point (2.6 G madd/s vs 2.0 at 1 wave and 1.5 at 4 waves).
template <class F> struct AccumWaves { static constexpr int value = 1; };
template <> struct AccumWaves<Fp2> { static constexpr int value = 2; };
template <class F>
__global__ void __launch_bounds__(64, AccumWaves<F>::value) k_msm_accum_affine(const Affine<F> *pts, const u32 *sorted, const u32 *start, const u32 *cnt,
                                                         const u32 *items, const u32 *item_start, u32 nkeys, u32 L,
                                                         XYZZ<F> *bucket, XYZZ<F> *partial_out) {
    // grid-stride over the items: the grid is sized from a host-side bound, the real count lives on the device
    const u32 total = item_start[nkeys], stride = gridDim.x * blockDim.x;
    for (u32 item = blockIdx.x * blockDim.x + threadIdx.x; item < total; item += stride)
        msm_accum_affine_body<F>(pts, sorted, start, cnt, items, item_start, nkeys, L, bucket, partial_out, item);
}
template <class F>
__global__ void __launch_bounds__(64, AccumWaves<F>::value) k_msm_accum_xyzz(const XYZZ<F> *partial_in, const u32 *start, const u32 *cnt, const u32 *items,
                                                       const u32 *item_start, u32 nkeys, u32 L, XYZZ<F> *bucket, XYZZ<F> *partial_out) {
    const u32 total = item_start[nkeys], stride = gridDim.x * blockDim.x;
    for (u32 item = blockIdx.x * blockDim.x + threadIdx.x; item < total; item += stride)
        msm_accum_xyzz_body<F>(partial_in, start, cnt, items, item_start, nkeys, L, bucket, partial_out, item);
}
// (G1: three waves per SIMD = 168 VGPRs, the slot a retired level-1 workgroup leaves; the compiler's own choice was 170)
template <class F> struct ReduceWaves { static constexpr int value = 3; };
template <> struct ReduceWaves<Fp2> { static constexpr int value = 1; };
template <class F>
__global__ void __launch_bounds__(64, ReduceWaves<F>::value) k_msm_bucket_reduce(const XYZZ<F> *bucket, u32 nbuckets, u32 seg, u32 tb, XYZZ<F> *out) {
    u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < tb) msm_bucket_reduce_body<F>(bucket, nbuckets, seg, out, blockIdx.y, t);
}


template <class F>
static void launch_accum_affine(hipStream_t st, unsigned grid, const void *pts, const u32 *sorted, const u32 *start, const u32 *cnt, const u32 *items,
                                const u32 *item_start, u32 nkeys, u32 L, void *bucket, void *pout) {
    hipLaunchKernelGGL(k_msm_accum_affine<F>, dim3(grid), dim3(64), 0, st, (const Affine<F> *)pts, sorted, start, cnt, items, item_start, nkeys, L,
                       (XYZZ<F> *)bucket, (XYZZ<F> *)pout);
}
template <class F>
static void launch_accum_xyzz(hipStream_t st, unsigned grid, const void *pin, const u32 *start, const u32 *cnt, const u32 *items, const u32 *item_start,
                              u32 nkeys, u32 L, void *bucket, void *pout) {
    hipLaunchKernelGGL(k_msm_accum_xyzz<F>, dim3(grid), dim3(64), 0, st, (const XYZZ<F> *)pin, start, cnt, items, item_start, nkeys, L,
                       (XYZZ<F> *)bucket, (XYZZ<F> *)pout);
}
template <class F>
static void launch_bucket_reduce(hipStream_t st, unsigned grid_x, unsigned nwin, const void *bucket, u32 nbuckets, u32 seg, u32 tb, void *out) {
    hipLaunchKernelGGL(k_msm_bucket_reduce<F>, dim3(grid_x, nwin), dim3(64), 0, st, (const XYZZ<F> *)bucket, nbuckets, seg, tb, (XYZZ<F> *)out);
}
// Sum of the bucket-reduce partials of every window: one workgroup adds SumT<F> consecutive points by a tree in LDS
// (log2 steps of one addition) instead of the item / level machinery's six launches of eight sequential additions each.
template <class F> struct SumT { static constexpr int value = 256; };
template <> struct SumT<Fp2> { static constexpr int value = 64; };   // the G2 addition wants > 256 VGPRs: one wave per workgroup
template <class F>
__global__ void __launch_bounds__(SumT<F>::value) k_msm_sum_tree(const XYZZ<F> *in, u32 n, u32 nout, XYZZ<F> *out) {
    extern __shared__ unsigned char sum_tree_lds[];
    XYZZ<F> *sh = reinterpret_cast<XYZZ<F> *>(sum_tree_lds);
    constexpr u32 T = SumT<F>::value;
    const u32 i = blockIdx.x * T + threadIdx.x, w = blockIdx.y;
    sh[threadIdx.x] = i < n ? in[(size_t)w * n + i] : XYZZ<F>::inf();
    __syncthreads();
    for (u32 h = T / 2; h > 0; h >>= 1) {
        if (threadIdx.x < h) {
            XYZZ<F> a = sh[threadIdx.x];
            xyzz_add(a, sh[threadIdx.x + h]);
            sh[threadIdx.x] = a;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) out[(size_t)w * nout + blockIdx.x] = sh[0];
}
template <class F>
static void launch_sum_tree(hipStream_t st, unsigned nout, unsigned nwin, const void *in, u32 n, void *out) {
    hipLaunchKernelGGL(k_msm_sum_tree<F>, dim3(nout, nwin), dim3(SumT<F>::value), SumT<F>::value * sizeof(XYZZ<F>), st, (const XYZZ<F> *)in, n, nout,
                       (XYZZ<F> *)out);
}
// item -> (key, first entry, end entry, "this item is its key's only one"): the 19-step binary search over item_start and the four
// dependent loads behind it, done once by a cheap, fully occupied kernel instead of at the head of every item of the heavy one
// (which runs 2 waves per SIMD and cannot hide that chain).  16 B per item, read back as one coalesced load.
template <class F>   // F only keeps the two translation units' instances apart
__global__ void __launch_bounds__(256) k_msm_item_table(const u32 *start, const u32 *cnt, const u32 *items, const u32 *item_start, u32 nkeys, uint4 *tab) {
    const u32 total = item_start[nkeys], stride = gridDim.x * blockDim.x;
    for (u32 item = blockIdx.x * blockDim.x + threadIdx.x; item < total; item += stride) {
        u32 key = msm_item_key(item_start, nkeys, item), b, e;
        msm_item_range(start[key], cnt[key], items[key], item - item_start[key], b, e);
        tab[item] = make_uint4(key, b, e, items[key] == 1 ? 1u : 0u);
    }
}
// ---- the finisher: ONE launch that ends the item machinery.  When the fullest key is down to <= a few thousand partial sums, every key
// that still holds more than one is on one of two device-side lists (k_msm_finish_list, msm.hip): keys with <= MSM_FIN_SMALL partial
// sums are summed by one thread each (the first nb_small workgroups, grid-stride over the small list); the others by one workgroup
// each: thread t adds the partial sums t, t + T, ... in place, then a tree over the first min(T, count) slots -- through the partial-sum
// array itself (global memory, workgroup barriers), so the same load / add / store forms serve every representation.  Replaces
// log_L2(count) more levels of three launches each on the tail of every MSM (the WHIR witness: one bucket of ~2 M entries, 255 of ~8 K).
// Form: per-thread running sum with load(p) / add(p) / store(p) over the partial-sum representation and to_bucket(b) (standard XYZZ).
template <class F> struct FinStd {
    typedef XYZZ<F> Partial;
    static constexpr int LDS_WORDS_PER_WAVE = 0;
    XYZZ<F> acc;
    MI_D explicit FinStd(u32 *) {}
    MI_D void load(const Partial *p) { acc = *p; }
    MI_D void add(const Partial *p) { xyzz_add(acc, *p); }
    MI_D void store(Partial *p) const { *p = acc; }
    MI_D void to_bucket(XYZZ<F> *b) const { *b = acc; }
};
template <class Form, class BucketT, int T, int WPS>
__global__ void __launch_bounds__(T, WPS) k_msm_finish_keys(typename Form::Partial *partials, const u32 *list_small, const u32 *list_big, const u32 *counters,
                                                           const u32 *item_start, const u32 *items, u32 nb_small, BucketT *bucket) {
    __shared__ u32 lds[Form::LDS_WORDS_PER_WAVE ? Form::LDS_WORDS_PER_WAVE * (T / 64) : 1];
    Form f(&lds[Form::LDS_WORDS_PER_WAVE ? (threadIdx.x >> 6) * Form::LDS_WORDS_PER_WAVE + (threadIdx.x & 63) : 0]);
    const u32 n_small = counters[0], n_big = counters[1], tid = threadIdx.x;
    if (blockIdx.x < nb_small) {
        for (u32 i = blockIdx.x * T + tid; i < n_small; i += nb_small * T) {
            const u32 key = list_small[i], base = item_start[key], cnt = items[key];
            f.load(partials + base);
            for (u32 k = 1; k < cnt; k++) f.add(partials + base + k);
            f.to_bucket(bucket + key);
        }
        return;
    }
    for (u32 i = blockIdx.x - nb_small; i < n_big; i += gridDim.x - nb_small) {   // (key, count) are the workgroup's: every barrier below is reached by all
        const u32 key = list_big[i], base = item_start[key], cnt = items[key];
        typename Form::Partial *slot = partials + base;
        u32 w = cnt < (u32)T ? cnt : (u32)T;   // live slots
        if (tid < w) {
            f.load(slot + tid);
            for (u32 k = tid + T; k < cnt; k += T) f.add(slot + k);
            f.store(slot + tid);
        }
        __syncthreads();
        u32 h = 1;
        while (h < w) h <<= 1;
        for (h >>= 1; h > 0; h >>= 1) {
            if (tid < h && tid + h < w) { f.add(slot + tid + h); f.store(slot + tid); }
            __syncthreads();
            w = w < h ? w : h;
        }
        if (tid == 0) f.to_bucket(bucket + key);
    }
}
template <class Form, class BucketT, int T, int WPS>
static void launch_finish_form(hipStream_t st, unsigned nb_small, unsigned nb_big, void *partials, const u32 *list_small, const u32 *list_big, const u32 *counters,
                               const u32 *item_start, const u32 *items, void *bucket) {
    hipLaunchKernelGGL((k_msm_finish_keys<Form, BucketT, T, WPS>), dim3(nb_small + nb_big), dim3(T), 0, st, (typename Form::Partial *)partials, list_small, list_big,
                       counters, item_start, items, nb_small, (BucketT *)bucket);
}
// Point-sharded MSM, SURVEY 8e option ii: own[i] += sum_p recv[p * own_len + i] -- the bucket sums the other devices hold for
// the keys this device owns, added to its own before the bucket reduce.
template <class F>
__global__ void __launch_bounds__(64) k_msm_sum_slices(XYZZ<F> *own, const XYZZ<F> *recv, u32 n_peers, u32 own_len) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= own_len) return;
    XYZZ<F> acc = own[i];
    for (u32 p = 0; p < n_peers; p++) xyzz_add(acc, recv[(size_t)p * own_len + i]);
    own[i] = acc;
}
template <class F>
static void launch_sum_slices(hipStream_t st, void *own, const void *recv, u32 n_peers, u32 own_len) {
    if (own_len) hipLaunchKernelGGL(k_msm_sum_slices<F>, dim3((own_len + 63) / 64), dim3(64), 0, st, (XYZZ<F> *)own, (const XYZZ<F> *)recv, n_peers, own_len);
}
template <class F>
__global__ void __launch_bounds__(64) k_msm2_precompute(const Affine<F> *base, Affine<F> *pre, u32 n, u32 c, u32 nwin) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) msm2_precompute_body<F>(base, pre, n, c, nwin, i);
}
template <class F>
static void launch_precompute(hipStream_t st, const void *base, void *pre, u32 n, u32 c, u32 nwin) {
    hipLaunchKernelGGL(k_msm2_precompute<F>, dim3((n + 63) / 64), dim3(64), 0, st, (const Affine<F> *)base, (Affine<F> *)pre, n, c, nwin);
}
template <class F>
static void host_combine_windows(const void *wsum, u32 nwin, u32 c, void *out) {
    *(XYZZ<F> *)out = nwin ? msm_combine_windows<F>((const XYZZ<F> *)wsum, nwin, c) : XYZZ<F>::inf();
}
