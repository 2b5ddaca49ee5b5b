/* mi355x_groth16_debug.h -- NOT part of the surface a maintainer binds: synthetic-input generators, throughput probes, plan / schedule
 * knobs, counters, tracing and fault injection of libmi355x_groth16.so.  For tests/, bench.py and tools/ only (the product
 * header no longer carries the lab bench).  Nothing here changes a result. */
#ifndef MI355X_GROTH16_DEBUG_H
#define MI355X_GROTH16_DEBUG_H
#include "mi355x_groth16.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- device-side test / bench utilities (not part of the reference surface) ---- */
#define MI_DIST_UNIFORM 0
#define MI_DIST_WHIR 1      /* 45% {0,1}, 25% bytes, 5% 64-bit, 25% uniform */
/* any other mix: per-mille shares of {0,1} / bytes / 64-bit values, the rest uniform Fr (each share <= 1000, their sum <= 1000).
 * tools/wire_census.py derives the shares the reference's circuit implies (profiles/r06_wire_census.txt). */
#define MI_DIST_MIX_FLAG 0x40000000
#define MI_DIST_MIX(bit_pm, byte_pm, u64_pm) (MI_DIST_MIX_FLAG | ((bit_pm) << 20) | ((byte_pm) << 10) | (u64_pm))
int32_t mi_gen_scalars_dev(mi_ctx *ctx, mi_fr *out_dev, size_t n, uint64_t seed, int dist);
int32_t mi_gen_g1_dev(mi_ctx *ctx, mi_g1_affine *out_dev, size_t n, uint64_t seed);
int32_t mi_gen_g2_dev(mi_ctx *ctx, mi_g2_affine *out_dev, size_t n, uint64_t seed);
/* elementwise field ops for parity tests of the device field layer:
 * field: 0 = Fr, 1 = Fp; op: 0 add, 1 sub, 2 mul, 3 inv(x), 4 to_mont(x), 5 from_mont(x),
 * 6 (xy + yx)/R and 7 (xy - yy)/R through the dual-product multiplier, 8 x^2 */
int32_t mi_field_op_dev(mi_ctx *ctx, int field, int op, void *z_dev, const void *x_dev,
                        const void *y_dev, size_t n);
/* out[i] = a[i] + b[i] on G1 (affine in, affine out; exercises add/double/inf cases) */
int32_t mi_g1_add_dev(mi_ctx *ctx, mi_g1_affine *out_dev, const mi_g1_affine *a_dev,
                      const mi_g1_affine *b_dev, size_t n);
int32_t mi_g2_add_dev(mi_ctx *ctx, mi_g2_affine *out_dev, const mi_g2_affine *a_dev,
                      const mi_g2_affine *b_dev, size_t n);
/* random-gather throughput probe: n_threads lanes each chain `iters` dependent 64-byte gathers from a table of n_entries
 * (a power of two) 64-byte entries; scratch: 1 KiB.  The ceiling the level-1 bucket accumulation's point gathers run against. */
int32_t mi_bench_gather_dev(mi_ctx *ctx, const void *table_dev, size_t n_entries, size_t n_threads, uint32_t iters,
                            void *scratch_dev, float *ms_out);
/* modular-multiply throughput probe: chains `iters` dependent Fp products per thread */
int32_t mi_bench_modmul_dev(mi_ctx *ctx, int field, size_t n_threads, uint32_t iters,
                            void *scratch_dev, float *ms_out);
/* raw VALU issue-rate probe (the integer-MAC ceiling SURVEY 8d asks to report beside the MSM):
 * kind 0 = 32x32+64 mad, 1 = mul_lo+mul_hi u32, 2 = fma f64, 3 = 24-bit mul+add, 4 = 64-bit add;
 * each thread runs 8 independent chains x iters steps */
int32_t mi_bench_valu_dev(mi_ctx *ctx, int kind, size_t n_threads, uint32_t iters,
                          void *scratch_dev, float *ms_out);
/* tuning / test knobs (0 = automatic).  NTT: tile = 2^log_e elements, radix caps of the contiguous and the
 * strided passes, threads per workgroup.  MSM: window bits c (2..16), item sizes of level 1 / later levels,
 * bucket-reduce segment, slices per window.  Tests use them to force multi-pass / multi-level paths at small n. */
int32_t mi_debug_set_ntt_plan(mi_ctx *ctx, uint32_t log_e, uint32_t max_contig, uint32_t max_strided);
int32_t mi_debug_set_ntt_threads(mi_ctx *ctx, uint32_t threads);
/* on = 1 (default): passes of radix >= 2^7 run seven of their stages in registers (wavefront butterflies); 0: every stage through
 * LDS.  direct_min_log_n: computeH builds its data-layout twiddle / coset tables from this size on (default 12; 29 = never). */
int32_t mi_debug_set_ntt_wave_stages(mi_ctx *ctx, uint32_t on, uint32_t direct_min_log_n);
/* Fused launches of computeH, a bit mask (default 7 = all).  Bit 0: the contiguous last pass of FFTInverse(a | b) and the contiguous
 * first pass of the coset FFT that follows run as one launch on the same tiles (csrc/ntt.hip k_ntt_contig_pair).  Bit 1: the strided
 * last pass of the coset FFT of a and of b, the product a b and the strided first pass of the last transform run as one launch
 * (k_ntt_strided_triple; plans whose first radix is 2^7 or 2^8).  Bit 2: the last pass of den FFTInverse(c) and the last pass of the last
 * transform, which subtracts it, run as one launch (k_ntt_contig_last_sub).  Same h whatever the mask; parity tests run the combinations. */
int32_t mi_debug_set_ntt_fuse_pair(mi_ctx *ctx, uint32_t on);
int32_t mi_debug_set_msm_plan(mi_ctx *ctx, uint32_t c, uint32_t L1, uint32_t L2, uint32_t seg, uint32_t G);
int32_t mi_debug_set_msm_chunk(mi_ctx *ctx, uint32_t chunk);   /* fixed-base sort: entries per pass-2 chunk */
int32_t mi_debug_set_msm_group_bits(mi_ctx *ctx, uint32_t gbits);   /* fixed-base sort: log2 buckets per pass-1 group, 6..15 */
/* window widths of the fixed-base tables the NEXT mi_pk_load[_dev] on ctx builds for the MSM groups A+K, B1+B2, Z:
 * 0 = automatic (tables when the MSM has >= 2^20 points and they fit in a third of the free device memory),
 * 1 = never, 17..22 = that width whatever the size */
int32_t mi_debug_set_prove_fixed_base(mi_ctx *ctx, uint32_t c_ak, uint32_t c_b, uint32_t c_z);
/* hold_accum = 1: inside a prove whose inputs are in HBM the wire MSMs (A, B1, B2, K) sort at once but start their bucket
 * accumulations only when computeH is done; 0 (default): everything as soon as its inputs exist.  Same proofs.  Measured neutral on
 * throughput and 0.4 ms worse on the single-proof latency (a proof alone is work-bound, not schedule-bound: DESIGN.md 7b). */
int32_t mi_debug_set_prove_schedule(mi_ctx *ctx, uint32_t hold_accum);
/* on = 0 (default): an MSM runs as many item levels as the fullest bucket of its sort needs (one 4-byte read-back per sort, waited for on the
 * thread that enqueues the accumulation); 1: as many as the worst case would (every entry in one bucket).  Same sums; parity tests run both. */
int32_t mi_debug_set_msm_bound_levels(mi_ctx *ctx, uint32_t on);
/* 1: generic MSMs of >= 2^18 pairs keep the one-pass counting sort instead of the LDS-staged two-pass one (parity tests run both) */
int32_t mi_debug_set_msm_one_pass_sort(mi_ctx *ctx, uint32_t on);
/* on = 1 (default): the G1 level-1 bucket accumulation runs in nine 29-bit limbs (keys loaded afterwards keep their G1 points in
 * the matching packed form) and the G1 partial sums between the levels stay in that form; 2: the same level 1 with standard-form
 * partial sums; 0: the 8 x 32-bit kernels everywhere.  Set before mi_pk_load; parity tests run all three. */
int32_t mi_debug_set_msm_limb29(mi_ctx *ctx, uint32_t on);
/* 3 (default) or 2: which build of the G1 level-1 29-bit kernel runs -- three waves per SIMD (fastest alone, shortest launches) or two
 * (leaves registers for other kernels' waves on the same SIMD: +1.7 % proofs/s with three proofs in flight at N = 2^23, slower when
 * one proof fills the GPU).  Same results. */
int32_t mi_debug_set_msm_l1_waves(mi_ctx *ctx, uint32_t waves);
/* EXPERIMENT, default 0 (off).  rounds = 1..4: the G1 level-1 accumulation by batch-affine rounds (affine additions, one shared inversion
 * per 64 * K additions; csrc/msm_ba_g1.cuh) wherever the buckets hold >= 32 entries on average and the scratch (768 B per item) fits.
 * 32 % fewer multiplications per addition, the same results -- and half the speed on MI355X: every pass is bound by its random 64-byte
 * reads at ~3 TB/s.  Kept for the measurement and the parity test. */
int32_t mi_debug_set_msm_batch_affine(mi_ctx *ctx, uint32_t rounds);
/* on = 1 (default): the fixed-base window tables of mi_pk_load / mi_msm_precompute_* convert to affine with one inversion per 16 points
 * (needs n XYZZ + n coordinates of scratch while building; falls back by itself without room); 0: one inversion per point.  Same tables. */
int32_t mi_debug_set_msm_precompute_batched(mi_ctx *ctx, uint32_t on);
/* Named measurement / test knobs of one context (the switches that are not worth an entry point each; none changes a result).
 * MI_EINVAL for an unknown name or a value out of range.  The library reads NO environment variable for any of this: the only
 * variables it looks at are MI_GROUP_TIMEOUT_MS and MI_GROUP_SHM_CHUNK_KB of the device groups (documented at mi_group_create_rank_ex).
 *   "l1_wg" 1 | 2 | 4        waves per workgroup of the G1 level-1 bucket-accumulate kernel, default 4 (a workgroup takes one slot on each
 *                            SIMD of a CU and returns them together, so the other streams' multi-wave workgroups find room)
 *   "g2_wg" 1 | 2 | 4        the same for the G2 level-1 kernel (each wave has its own 18 KiB LDS accumulator image)
 *   "l1_waves" 2 | 3         = mi_debug_set_msm_l1_waves
 *   "z_waves" 0 | 2          2: the Z MSM's level-1 launch alone on the two-waves-per-SIMD build
 *   "g1_grid_per_cu", "g2_grid_per_cu"   resident-grid cap per CU of the level-1 launches, in waves (0 = 128)
 *   "count_per" 0..64        fixed-base sort: slices per counting workgroup (0 = 32)
 *   "plain_scatter" 0 | 1    fixed-base sort: pass 2 by the plain scatter instead of the LDS-staged one
 *   "finisher" 0 | 1         1 (default): once no bucket holds more than "finisher_max" partial sums the item levels end in ONE launch
 *                            (k_msm_finish_keys) instead of log_8 more levels of three launches each
 *   "z_count_fused" 0 | 1     1 (default): inside a proof the Z MSM's sort takes its digit count from computeH's last launch (the kernel that
 *                            stores h counts the digits of what it stores: h is read once less) instead of a count pass of its own.  Same
 *                            proofs; throughput equal (the count's instructions move, they do not go away), one proof alone 0.1-0.2 ms shorter
 *   "flat_item_l1" 0 | 1 | 4..64   entries per level-1 item of a FLAT sort (fullest bucket <= 2 x the average: uniform scalars, e.g. the h
 *                            coefficients of a proof's Z MSM).  0 (default) = automatic: average / L2^k where that falls into 17..32 (26 at
 *                            N = 2^23), so that the levels above are full L2-ary trees; 1 = off (the plan's L1); 4..64 = forced
 *   "dense_item_l1" 0 | 1 | 4..64   entries per level-1 item of a DENSE sort that is not flat (>= half of the n x windows digits non-zero: the wire
 *                            values of a witness of mostly full-width field elements, what the reference's circuit implies).  0 (default) = automatic
 *                            (32: half the partial sums for the upper levels); 1 = off (the plan's L1 = 16); 4..64 = forced
 *   "finisher_max" 0..2^20   0 = automatic (G1 4096, G2 1024)
 *   "finisher_min_level" 0..16   the finisher follows accumulate pass number this + 1 at the earliest (default 2: the first two passes
 *                            are where every ordinary bucket ends; a finisher over 2^19 buckets of 13 partial sums each measured -5 %)
 *   "item_l1", "item_l2", "reduce_seg"   = the L1, L2, seg of mi_debug_set_msm_plan, one at a time (a flat sort keeps its own item size: "flat_item_l1")
 *   "hold_accum" 0 | 1       = mi_debug_set_prove_schedule
 *   "ntt_lds_floor_kb" 0..160   LDS every NTT pass workgroup requests at least (caps the workgroups per CU) */
int32_t mi_debug_set_knob(mi_ctx *ctx, const char *name, int64_t value);
/* Counters the tests read to prove that an optional path really ran: "z_count_fused_launches" = computeH last launches of this context that
 * carried the Z MSM's digit count (knob "z_count_fused"); "dense_item_sorts" = bucket accumulations whose level-1 item size the dense-sort rule
 * chose (knob "dense_item_l1").  MI_EINVAL for an unknown name. */
int32_t mi_debug_get_counter(mi_ctx *ctx, const char *name, uint64_t *out);
/* Process-wide, for contexts created afterwards: how the MSM slots of a context share streams (0: K's stream created, destroyed and
 * pointed at B1's, as rounds 4-5 did; 1, the default: never created; 2: A, B1 and K on one stream).  Same results; an experiment on which
 * chains end up on one hardware queue. */
int32_t mi_debug_set_stream_plan(int32_t plan);
/* the old name of mi_set_trace_ranges (mi355x_groth16.h) */
int32_t mi_debug_set_trace_ranges(int32_t on);
/* error-path tests: the nth MI-checked HIP call from now (library-wide, any thread) fails with hipErrorUnknown instead of
 * running; 0 disarms.  Used to prove that init / load / prove unwind without leaks or crashes. */
int32_t mi_debug_inject_hip_failure(int32_t nth);

#ifdef __cplusplus
}
#endif
#endif
