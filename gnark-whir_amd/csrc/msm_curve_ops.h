// Per-curve entry points of the MSM (kernel launchers + the host-side window combine).  msm.hip holds the curve-
// independent orchestration (sort, item levels, streams); msm_g1.hip / msm_g2.hip instantiate the point kernels, so the
// two heavy translation units build in parallel.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

static constexpr uint32_t MSM_FIN_SMALL = 16;   // the finisher: keys with <= this many partial sums are summed by one thread each
struct MsmCurveOps {
    size_t xyzz_bytes;   // sizeof(XYZZ<F>): 128 (G1) / 256 (G2)
    void (*accum_affine)(hipStream_t st, unsigned grid, const void *pts, const uint32_t *sorted, const uint32_t *start, const uint32_t *cnt,
                         const uint32_t *items, const uint32_t *item_start, uint32_t nkeys, uint32_t L, void *bucket, void *partial_out);
    void (*accum_xyzz)(hipStream_t st, unsigned grid, const void *partial_in, const uint32_t *start, const uint32_t *cnt, const uint32_t *items,
                       const uint32_t *item_start, uint32_t nkeys, uint32_t L, void *bucket, void *partial_out);
    void (*bucket_reduce)(hipStream_t st, unsigned grid_x, unsigned nwin, const void *bucket, uint32_t nbuckets, uint32_t seg, uint32_t tb, void *out);
    // tree sum: block (bx, w) adds in[w*n + bx*sum_T .. + sum_T) in LDS and writes out[w*nout + bx]
    uint32_t sum_T;
    void (*sum_tree)(hipStream_t st, unsigned nout, unsigned nwin, const void *in, uint32_t n, void *out);
    void (*precompute)(hipStream_t st, const void *base, void *pre, uint32_t n, uint32_t c, uint32_t nwin);   // fixed-base window copies
    // the same copies with one inversion per 16 points instead of one per point (batch_affine.cuh); state: n XYZZ, prefix: n coordinates of scratch
    void (*precompute_batched)(hipStream_t st, const void *base, void *pre, uint32_t n, uint32_t c, uint32_t nwin, void *state, void *prefix);
    size_t coord_bytes;   // sizeof(F): 32 (G1) / 64 (G2)
    // total = sum_w 2^(c*w) * wsum[w] on the host; nwin == 0 yields the point at infinity
    void (*combine_windows)(const void *host_wsum, uint32_t nwin, uint32_t c, void *out_xyzz);
    // own[i] += sum_p recv[p * own_len + i]  (XYZZ; bucket slices received from the other devices of a sharded MSM)
    void (*sum_slices)(hipStream_t st, void *own, const void *recv, uint32_t n_peers, uint32_t own_len);
    // Level-1 accumulation over points kept in the R' = 2^261 packed form (curve29.cuh: nine 29-bit limbs, lazy arithmetic);
    // same arguments and results as accum_affine.
    void (*accum_affine_rp)(hipStream_t st, unsigned grid, const void *pts_rp, const uint32_t *sorted, const uint32_t *start, const uint32_t *cnt,
                            const uint32_t *items, const uint32_t *item_start, uint32_t nkeys, uint32_t L, void *bucket, void *partial_out,
                            void *item_table /* 16 B per item of scratch */, uint32_t rp_partials /* bit 0; bit 1: the G1 kernel's two-wave build */,
                            hipEvent_t ev_before /* recorded on st between the item-table kernel and the accumulate kernel when non-null (stats) */);
    // Levels >= 2 over partial sums the level before left in the packed R' form (accum_affine_rp with rp_partials = 1, or this
    // kernel): same arguments as accum_xyzz; bucket sums leave in the standard form, partial sums in the R' form.  Null = the
    // curve keeps its partial sums in the standard form, and rp_partials must be 0.  (G1: k_msm_accum_xyzz29, G2: k_msm_accum_xyzz_g2_29.)
    void (*accum_xyzz_rp)(hipStream_t st, unsigned grid, const void *partial_in, const uint32_t *start, const uint32_t *cnt, const uint32_t *items,
                          const uint32_t *item_start, uint32_t nkeys, uint32_t L, void *bucket, void *partial_out);
    // Level-1 accumulation by batch-affine rounds (msm_ba_g1.cuh): same inputs and outputs as accum_affine_rp for items of <= 16
    // entries.  rounds = 1..4; scratch: nodes (512 B per item), prefix (256 B per item), totals / invs (msm_ba_scratch_bytes each).
    // Null where no such kernels exist (G2).
    void (*accum_affine_ba)(hipStream_t st, unsigned grid_cap, const void *pts_rp, const uint32_t *sorted, const uint32_t *start, const uint32_t *cnt,
                            const uint32_t *items, const uint32_t *item_start, uint32_t nkeys, void *bucket, void *partial_out, void *item_table,
                            uint32_t rp_partials, uint32_t rounds, uint64_t items_bound, uint32_t target_waves, void *nodes, void *prefix, void *totals,
                            void *invs, hipEvent_t ev_before);
    // The finisher (msm_curve_kernels.cuh k_msm_finish_keys): every key on the two lists (k_msm_finish_list) has its partial sums
    // partials[item_start[key] .. + items[key]) -- standard XYZZ, or the packed R' form when rp -- summed into bucket[key] (standard).
    // Workgroups of finish_T threads; a key on the big list may hold at most finish_max partial sums (the chain of its one workgroup).
    void (*finish_keys)(hipStream_t st, unsigned nb_small, unsigned nb_big, void *partials, const uint32_t *list_small, const uint32_t *list_big,
                        const uint32_t *counters, const uint32_t *item_start, const uint32_t *items, void *bucket, uint32_t rp);
    uint32_t finish_T, finish_max;
    // dst[i] = src[i] with both coordinates multiplied by 2^5 mod p: standard Montgomery form -> the R' packed form (dst may be src)
    void (*to_rprime)(hipStream_t st, void *dst, const void *src, size_t n);
};
// slots per lane and chunk (K) of a batch-affine round over at most `slots_bound` slots: enough chunks to fill `target_waves` waves, 4..32
static inline uint32_t msm_ba_K(uint64_t slots_bound, uint32_t target_waves) {
    const uint64_t k = slots_bound / (64ull * (target_waves ? target_waves : 1));
    return k < 4 ? 4u : k > 32 ? 32u : (uint32_t)k;
}
static inline uint64_t msm_ba_chunks(uint64_t slots_bound, uint32_t target_waves) {
    const uint64_t per = 64ull * msm_ba_K(slots_bound, target_waves);
    return (slots_bound + per - 1) / per;
}
// bytes of the totals (and of the invs) scratch array: the largest round is the first (8 slots per item)
static inline size_t msm_ba_scratch_bytes(uint64_t items_bound, uint32_t target_waves) {
    size_t m = 0;
    for (int r = 1; r <= 4; r++) {
        const size_t b = (size_t)msm_ba_chunks(items_bound << (4 - r), target_waves) * 64 * 32;
        m = b > m ? b : m;
    }
    return m;
}
const MsmCurveOps &msm_g1_ops();   // msm_g1.hip
const MsmCurveOps &msm_g2_ops();   // msm_g2.hip
