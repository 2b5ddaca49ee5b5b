/* mi355x_groth16.h — C-ABI of the MI355X-native Groth16 prove path (BN254).
 *
 * Drop-in boundary for the one hot path of reilabs/gnark-whir: everything that happens inside
 *     proof, _ := groth16.Prove(ccs, pk, witness, backend.WithSolverOptions(...))
 * at /root/reference/mt.go:496 *after* gnark's R1CS solver has produced the wire vector W
 * and the per-constraint vectors a, b, c — i.e. computeH (gnark's 7 NTTs, done with 6), the scalar filters, four
 * G1 MSMs, one G2 MSM and proof assembly (SURVEY.md section 3.3 steps 4-8, section 8a rows
 * a3-a9).  The reference has no FFI of its own; the seam a maintainer binds is gnark's
 * accelerator seam (backend/groth16/bn254/icicle in gnark v0.11.0, go.mod:6), whose shape
 * (device-resident proving key + one Prove call) these entry points mirror.  The cgo
 * binding a maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions
 *   - every function returns an int32 status (MI_OK == 0, negative on error); no C++
 *     exception crosses the ABI; mi_last_error() gives a message for the last failure.
 *   - mi_fr / mi_fp are 4 x uint64 little-endian limbs in Montgomery form (R = 2^256), the
 *     in-memory layout of gnark-crypto's fr.Element / fp.Element.  The limb order is the one
 *     /root/reference/typeConverters/typeConverters.go:30-39 spells out (that file's values
 *     are canonical, not Montgomery).  Go passes unsafe.Pointer(&slice[0]) with zero copies.
 *   - affine infinity is (0,0); Jacobian infinity has Z == 0 (gnark-crypto's encoding).
 *   - functions without suffix take HOST pointers and copy over PCIe; "_dev" variants take
 *     DEVICE pointers (hipMalloc'ed, 32-byte aligned) and never touch host memory.
 *   - the caller owns every buffer; the library keeps no caller pointer after return
 *     (cgo rule).  mi_ctx / mi_pk are opaque library-owned handles.
 *   - one mi_ctx drives one GPU.  Calls on one ctx must not overlap (no internal locking); inside a call the library
 *     fans work out over its own HIP streams and joins them before returning.
 */
#ifndef MI355X_GROTH16_H
#define MI355X_GROTH16_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes ---- */
#define MI_OK 0
#define MI_EINVAL (-1)   /* bad argument: size not a power of two, log_n > 28, null pointer */
#define MI_EHIP (-2)     /* a HIP runtime call failed; see mi_last_error */
#define MI_ENOMEM (-3)   /* device or host allocation failed */
#define MI_ENODEV (-4)   /* no usable gfx950 device / code object missing */

/* ---- value types (layout == gnark-crypto structs) ---- */
typedef struct { uint64_t l[4]; } mi_fr;              /* fr.Element, replaces ecc/bn254/fr */
typedef struct { uint64_t l[4]; } mi_fp;              /* fp.Element */
typedef struct { mi_fp a0, a1; } mi_fp2;              /* fptower.E2: a0 + a1*u, u^2 = -1 */
typedef struct { mi_fp x, y; } mi_g1_affine;          /* bn254.G1Affine, 64 B */
typedef struct { mi_fp x, y, z; } mi_g1_jac;          /* bn254.G1Jac, 96 B */
typedef struct { mi_fp2 x, y; } mi_g2_affine;         /* bn254.G2Affine, 128 B */
typedef struct { mi_fp2 x, y, z; } mi_g2_jac;         /* bn254.G2Jac, 192 B */

typedef struct mi_ctx mi_ctx;
typedef struct mi_pk mi_pk;

/* ---- NTT flags: mirror fft.Domain.FFT / FFTInverse options (gnark-crypto fr/fft) ---- */
#define MI_NTT_INVERSE 1u   /* FFTInverse: uses GeneratorInv and scales by CardinalityInv */
#define MI_NTT_COSET 2u     /* fft.OnCoset(): shift by FrMultiplicativeGen = 5 */
#define MI_NTT_DIT 4u       /* fft.DIT: bit-reversed in, natural out. Default fft.DIF:
                               natural in, bit-reversed out */

/* ---- MSM flags ---- */
#define MI_MSM_SCALARS_CANONICAL 1u /* scalars are plain integers < r (default: Montgomery) */

/* Proving key as gnark keeps it in memory (groth16/bn254 ProvingKey, built by groth16.Setup at
 * mt.go:448).  All arrays are read during mi_pk_load only. */
typedef struct mi_pk_desc {
    uint32_t log_n;                 /* pk.Domain.Cardinality = 2^log_n, log_n <= 28          */
    uint32_t nb_public;             /* r1cs.GetNbPublicVariables(), includes the ONE wire     */
    uint64_t nb_wires;              /* len(solution.W)                                        */
    const mi_g1_affine *g1_a;  uint64_t n_g1_a;   /* pk.G1.A, points at infinity filtered out */
    const mi_g1_affine *g1_b;  uint64_t n_g1_b;   /* pk.G1.B, idem                            */
    const mi_g1_affine *g1_k;  uint64_t n_g1_k;   /* pk.G1.K, private non-committed wires     */
    const mi_g1_affine *g1_z;  uint64_t n_g1_z;   /* pk.G1.Z, >= 2^log_n - 1 points, in the
                                                     bit-reversed order Setup leaves them in  */
    const mi_g2_affine *g2_b;  uint64_t n_g2_b;   /* pk.G2.B                                  */
    mi_g1_affine alpha1, beta1, delta1;           /* pk.G1.Alpha / Beta / Delta               */
    mi_g2_affine beta2, delta2;                   /* pk.G2.Beta / Delta                       */
    const uint8_t *infinity_a;      /* pk.InfinityA, nb_wires Go bools (1 byte each)          */
    const uint8_t *infinity_b;      /* pk.InfinityB                                           */
    const uint32_t *committed_wires;/* wires removed from the K MSM (BSB22 private committed +
                                       commitment wires), ascending; may be NULL              */
    uint64_t n_committed;
} mi_pk_desc;

/* groth16 Proof{Ar, Bs, Krs} exactly as gnark's struct holds it (affine, Montgomery). */
typedef struct mi_proof_out {
    mi_g1_affine ar;
    mi_g2_affine bs;
    mi_g1_affine krs;
} mi_proof_out;

/* Per-phase times in milliseconds.  The prove path runs computeH on the ctx stream and the five MSMs on five library
 * streams, so the compute_h / msm_* spans (HIP events on their own streams) OVERLAP and do not add up to total_ms.
 *   h2d_ms       host-pointer entry points only: upload of W, a, b, c
 *   filter_ms    host blinding work (r*delta, s*delta, s*Ar, r*Bs1 ...) done while the GPU is still busy
 *   assemble_ms  host work left after the last MSM landed (final additions, conversion to affine)
 *   total_ms     wall clock of the call (host clock) */
typedef struct mi_stats {
    float h2d_ms, compute_h_ms, filter_ms;
    float msm_a_ms, msm_b1_ms, msm_k_ms, msm_z_ms, msm_b2_ms;
    float assemble_ms, total_ms;
    /* dominant-kernel accounting for the roofline line (last prove / msm call) */
    float g1_accum_kernel_ms;       /* summed duration of the G1 bucket-accumulate launches   */
    uint64_t g1_accum_pairs;        /* (point, scalar) pairs those launches consumed          */
    uint32_t g1_accum_launches;
    float ntt_kernel_ms;            /* summed duration of NTT pass launches                   */
    uint64_t ntt_elems;             /* elements transformed (N per size-N transform)          */
    uint32_t ntt_launches;
    uint64_t g1_accum_entries;      /* non-zero digits = mixed additions those launches performed     */
    /* (a prove times the level-1 launches of A and Z only -- B1 and K share a stream, so a bracket around one may hold kernels of the
     *  other: the four g1_accum_* fields above cover those two launches) */
    uint64_t g1_level1_additions;   /* mixed additions of ALL G1 level-1 launches of the call (A, B1, K, Z), timed or not */
} mi_stats;

/* ---- lifecycle ---- */
int32_t mi_init(int device_id, mi_ctx **out);             /* replaces icicle device init      */
/* mi_init with an explicit stream-priority scheme (the device has three levels): 0 = a context on its own (computeH high,
 * the wire MSMs A/B1/B2/K normal, Z low: shortest single-proof latency) = mi_init; 1 / 2 / 3 = first / second / further
 * context of a group that proves concurrently on one GPU (1: all high but Z normal; 2: normal, Z low; 3: all low), so that
 * one proof runs nearly as if alone and the others fill what it leaves.  mi_prover_create staggers its contexts this way. */
int32_t mi_init_prio(int device_id, int prio_scheme, mi_ctx **out);
int32_t mi_shutdown(mi_ctx *ctx);
const char *mi_last_error(mi_ctx *ctx);                   /* never NULL                       */
/* Use the caller's HIP stream (a hipStream_t passed as void*) for all later work of ctx.  The context's own stream
 * is created at the device's highest priority (it carries computeH, the head of a proof's longest chain; the Z MSM runs
 * on a low-priority library stream so that the other MSMs' tails hide under it): pass a high-priority stream to keep that. */
int32_t mi_set_stream(mi_ctx *ctx, void *hip_stream);

/* ---- proving key: upload once, device-resident across proofs ---- */
int32_t mi_pk_load(mi_ctx *ctx, const mi_pk_desc *desc, mi_pk **out);      /* host arrays     */
int32_t mi_pk_load_dev(mi_ctx *ctx, const mi_pk_desc *desc, mi_pk **out);  /* point arrays are
        device pointers that the pk ADOPTS BY REFERENCE (caller keeps them alive); masks and
        committed_wires stay host pointers                                                    */
int32_t mi_pk_free(mi_ctx *ctx, mi_pk *pk);

/* ---- proving key from gnark's own serialisation: the stream groth16 bn254 ProvingKey.WriteRawTo writes
 * (gnark v0.11.0, go.mod:6; the reference itself re-runs Setup on every run, mt.go:448, and never writes one).
 * EXPERIMENTAL: LAYOUT RECALLED, UNVERIFIED -- spelled out in oracle/pk_raw.py and csrc/pk_raw.hip; the parser cross-checks every count
 * and refuses a shifted layout with MI_EINVAL, but the bit order inside the packed InfinityA/B masks (taken as MSB-first) is checked by
 * nothing: until a real gnark fixture has been loaded and a proof from it verified, do not rely on this entry point (mi_pk_load is the
 * supported one).  nb_public and the wires removed from the K MSM come from the constraint system, not
 * from the key file.  ped_out (room for MI_PK_RAW_MAX_COMMITMENTS handles, may be NULL) receives the key's Pedersen
 * commitment keys for mi_pedersen_*; free them with mi_pedersen_pk_free. ---- */
#define MI_PK_RAW_MAX_COMMITMENTS 16
typedef struct mi_pk_raw_info {
    uint32_t log_n, n_commitment_keys;
    uint64_t nb_wires, n_g1_a, n_g1_b, n_g1_z, n_g1_k, n_g2_b;
    uint64_t off_alpha1, off_g1_a, off_g1_b, off_g1_z, off_g1_k, off_beta2, off_g2_b, off_infinity_a, off_infinity_b;
    uint64_t n_basis[MI_PK_RAW_MAX_COMMITMENTS], off_basis[MI_PK_RAW_MAX_COMMITMENTS], off_basis_exp_sigma[MI_PK_RAW_MAX_COMMITMENTS];
} mi_pk_raw_info;
int32_t mi_pk_raw_inspect(const uint8_t *buf, size_t len, mi_pk_raw_info *info);   /* host only: section offsets and counts */
typedef struct mi_pedersen_pk mi_pedersen_pk;
int32_t mi_pk_load_raw(mi_ctx *ctx, const uint8_t *buf, size_t len, uint32_t nb_public, const uint32_t *committed_wires,
                       size_t n_committed, mi_pk **out, mi_pedersen_pk **ped_out, uint32_t *n_ped_out);

/* ---- fft.Domain.FFT / FFTInverse over Fr, in place, n = 2^log_n (row a3/a4) ---- */
int32_t mi_ntt(mi_ctx *ctx, mi_fr *inout, uint32_t log_n, uint32_t flags);
int32_t mi_ntt_dev(mi_ctx *ctx, mi_fr *inout_dev, uint32_t log_n, uint32_t flags);

/* ---- computeH (gnark prove.go): a, b, c have n_constraints entries, zero-padded to 2^log_n;
 * h_out receives 2^log_n elements in the bit-reversed order gnark leaves them in.  c == NULL: c = a o b (see mi_groth16_prove) ---- */
int32_t mi_compute_h(mi_ctx *ctx, uint32_t log_n, const mi_fr *a, const mi_fr *b, const mi_fr *c,
                     size_t n_constraints, mi_fr *h_out);
int32_t mi_compute_h_dev(mi_ctx *ctx, uint32_t log_n, const mi_fr *a_dev, const mi_fr *b_dev,
                         const mi_fr *c_dev, size_t n_constraints, mi_fr *h_out_dev);

/* ---- G1Jac.MultiExp / G2Jac.MultiExp (rows a5, a6, a8).  out is a HOST pointer in both
 * variants; the result is normalised (Z = 1, or X = Y = 1, Z = 0 for infinity). ---- */
int32_t mi_msm_g1(mi_ctx *ctx, const mi_g1_affine *pts, const mi_fr *scalars, size_t n,
                  uint32_t flags, mi_g1_jac *out);
int32_t mi_msm_g1_dev(mi_ctx *ctx, const mi_g1_affine *pts_dev, const mi_fr *scalars_dev, size_t n,
                      uint32_t flags, mi_g1_jac *out);
int32_t mi_msm_g2(mi_ctx *ctx, const mi_g2_affine *pts, const mi_fr *scalars, size_t n,
                  uint32_t flags, mi_g2_jac *out);
int32_t mi_msm_g2_dev(mi_ctx *ctx, const mi_g2_affine *pts_dev, const mi_fr *scalars_dev, size_t n,
                      uint32_t flags, mi_g2_jac *out);

/* ---- fixed-base MSM: when the bases are static (a proving key), store next to every base P_i the window copies
 * 2^(c*w) * P_i, w < ceil(256/c) (c in 17..22; pre holds ceil(256/c) * n points, [w][i] order).  All windows then share
 * one set of 2^(c-1) buckets and a 254-bit scalar costs ceil(256/c) = 12 mixed additions at c = 22 instead of 16.  Same
 * result as mi_msm_g1/g2.  mi_pk_load builds such tables for the prove path's large MSMs when they fit. ---- */
int32_t mi_msm_precompute_g1_dev(mi_ctx *ctx, const mi_g1_affine *base_dev, size_t n, uint32_t c, mi_g1_affine *pre_dev);
int32_t mi_msm_precompute_g2_dev(mi_ctx *ctx, const mi_g2_affine *base_dev, size_t n, uint32_t c, mi_g2_affine *pre_dev);
/* In place: the table's coordinates times 2^5 mod p -- the packed R' = 2^261 form the 9 x 29-bit level-1 kernels gather from (what
 * mi_pk_load keeps its own tables in).  A converted table is passed to mi_msm_g*_fixed_dev with MI_MSM_TABLE_RPRIME; it is no longer a
 * table of standard Montgomery points.  n_points = ceil(256/c) * n. */
int32_t mi_msm_table_to_rprime_g1_dev(mi_ctx *ctx, mi_g1_affine *pre_dev, size_t n_points);
int32_t mi_msm_table_to_rprime_g2_dev(mi_ctx *ctx, mi_g2_affine *pre_dev, size_t n_points);
#define MI_MSM_TABLE_RPRIME 2u /* flags of mi_msm_g*_fixed_dev: pre_dev went through mi_msm_table_to_rprime_* */
int32_t mi_msm_g1_fixed_dev(mi_ctx *ctx, const mi_g1_affine *pre_dev, const mi_fr *scalars_dev, size_t n, uint32_t c,
                            uint32_t flags, mi_g1_jac *out);
int32_t mi_msm_g2_fixed_dev(mi_ctx *ctx, const mi_g2_affine *pre_dev, const mi_fr *scalars_dev, size_t n, uint32_t c,
                            uint32_t flags, mi_g2_jac *out);

/* ---- the fused prove path: replaces groth16.Prove after the solve (mt.go:496).
 * W: nb_wires wire values; a, b, c: n_constraints values each (solution.A/B/C);
 * r, s: the two blinding scalars gnark samples with fr.SetRandom (passed in so that CPU and
 * GPU proofs of the same (pk, witness, r, s) are byte-identical — SURVEY section 7 H1).
 * stats may be NULL.
 * c may be NULL (every prove / submit / compute_h entry point): c is then formed on the device as a o b, row by row, on the way into
 * its transform -- what solution.C is for every witness gnark's solver accepts (the solver returns satisfied constraints only), so
 * the proof bytes are the same and a quarter of a host-input proof's PCIe bytes never crosses.  Pass c when a, b, c may NOT satisfy
 * a o b = c (gnark's computeH does not assume it, and neither does the general path). ---- */
int32_t mi_groth16_prove(mi_ctx *ctx, mi_pk *pk, const mi_fr *W, size_t n_wires,
                         const mi_fr *a, const mi_fr *b, const mi_fr *c, size_t n_constraints,
                         const mi_fr *r, const mi_fr *s, mi_proof_out *out, mi_stats *stats);
int32_t mi_groth16_prove_dev(mi_ctx *ctx, mi_pk *pk, const mi_fr *W_dev, size_t n_wires,
                             const mi_fr *a_dev, const mi_fr *b_dev, const mi_fr *c_dev,
                             size_t n_constraints, const mi_fr *r, const mi_fr *s,
                             mi_proof_out *out, mi_stats *stats);
/* Stats of the last mi_msm_* / mi_ntt* / mi_compute_h* / prove call on ctx. */
int32_t mi_get_stats(mi_ctx *ctx, mi_stats *out);
/* Device memory the library holds, in bytes: what a proving key keeps resident (pk may be NULL) and what a context has grown to
 * (grow-only workspaces; a prover pool has one such set per context in flight).  For capacity planning at N = 2^26. */
typedef struct mi_mem_ledger {
    uint64_t key_bases;        /* pk.G1.{A,B,K,Z} / pk.G2.B arrays the key still holds (incl. per-wire expanded and converted copies) */
    uint64_t key_tables;       /* fixed-base window tables */
    uint64_t key_indices;      /* gather-index arrays from the infinity masks */
    uint64_t ctx_ntt_tables;   /* computeH's twiddle / coset tables of this context */
    uint64_t ctx_ntt_vectors;  /* the two N-element vectors computeH works in + h */
    uint64_t ctx_msm;          /* workspaces of the context's MSM slots (digits, histograms, sorted entries, partial sums, buckets) */
    uint64_t ctx_other;        /* staging areas for host inputs and the rest */
} mi_mem_ledger;
int32_t mi_get_mem_ledger(mi_ctx *ctx, const mi_pk *pk, mi_mem_ledger *out);
/* window bits of the fixed-base tables the key was loaded with, for the MSM groups A+K, B1+B2, Z (0 = that group runs the generic plan) */
int32_t mi_pk_table_plan(const mi_pk *pk, uint32_t c_out[3]);
/* Workspaces only ever grow (no hipMalloc in steady state).  mi_ctx_trim gives them back: every scratch buffer, MSM slot array and NTT
 * table of an IDLE context is freed (streams, events and keys stay); the next call grows what it needs again.  For a service that has
 * proved an N = 2^26 circuit and goes back to 2^23, or before loading a second large key.  mi_prover_trim does the same for every
 * context and input set of an idle pool (MI_EINVAL while jobs are queued or running, or while a mi_prover_commit is in progress: commits
 * count as activity). */
int32_t mi_ctx_trim(mi_ctx *ctx);

/* ---- prover pool: several proofs in flight on one device.
 * The reference proves one circuit per groth16.Prove call (mt.go:496) and a prover service issues those calls from many
 * goroutines; gnark's CPU prover then shares the cores between them.  The drop-in equivalent: `in_flight` contexts on one
 * GPU, each with its own streams, workspaces and host worker thread, fed from one queue, so that the GPU fills the
 * serial head and tail of one proof (first NTT passes, last MSM's reduce, host assembly) with the bulk of another.
 * The proving key is read-only during prove: load it once with mi_pk_load[_dev] on mi_prover_ctx(p, 0) and share it.
 * submit returns at once with a ticket; W/a/b/c (host or device memory as the variant says), out and stats must stay
 * valid until mi_prover_wait(ticket) returns the job's status (r and s are copied).  Each ticket is waited on exactly
 * once, from any thread.  Proofs are bit-identical to mi_groth16_prove[_dev] on the same inputs.
 * Host inputs (mi_prover_submit) pass through an upload stage with in_flight + 1 device input sets; a job goes to an IDLE worker as soon
 * as W is resident (its A, B1, B2, K MSMs start) and a, b, c follow; otherwise when all four are.  Null W / a / b / c are accepted where the matching
 * count is 0.  The input sets are grow-only: the first job of a LARGER size than any before frees and reallocates its set (hipFree
 * synchronises the device -- a one-off stall of every proof in flight, like any other workspace growth).
 * mi_prover_destroy runs the jobs still queued, then frees every context. ---- */
typedef struct mi_prover mi_prover;
int32_t mi_prover_create(int device_id, uint32_t in_flight /* 1..16 */, mi_prover **out);
int32_t mi_prover_destroy(mi_prover *p);
uint32_t mi_prover_in_flight(const mi_prover *p);
mi_ctx *mi_prover_ctx(mi_prover *p, uint32_t i);        /* i < in_flight; NULL otherwise.  Prove on it yourself only while the pool is idle */
const char *mi_prover_last_error(mi_prover *p);         /* message of the last job whose mi_prover_wait returned != MI_OK */
int32_t mi_prover_submit(mi_prover *p, mi_pk *pk, const mi_fr *W, size_t n_wires,
                         const mi_fr *a, const mi_fr *b, const mi_fr *c, size_t n_constraints,
                         const mi_fr *r, const mi_fr *s, mi_proof_out *out, mi_stats *stats, uint64_t *ticket);
int32_t mi_prover_submit_dev(mi_prover *p, mi_pk *pk, const mi_fr *W_dev, size_t n_wires,
                             const mi_fr *a_dev, const mi_fr *b_dev, const mi_fr *c_dev, size_t n_constraints,
                             const mi_fr *r, const mi_fr *s, mi_proof_out *out, mi_stats *stats, uint64_t *ticket);
int32_t mi_prover_wait(mi_prover *p, uint64_t ticket);
int32_t mi_prover_trim(mi_prover *p);
/* The proof the WHIR circuit really produces carries one BSB22 commitment (/root/reference/utilities/utilities.go:189
 * logderivlookup.New and mtUtilities.go:452 uints.New force it: SURVEY 3.3 steps 1 and 3, row a10).  Through the pool:
 *   mi_prover_commit        pedersen Commit INSIDE the solve (the hint override): synchronous, callable from any thread, one
 *                           commitment at a time on a context of the pool's own; the SHA-256 hash-to-field stays in Go.
 *   mi_prover_submit_bsb22  mi_prover_submit (host inputs) whose job also computes the proof's CommitmentPok as prove.go does --
 *                           pedersen.BatchProve: sum_i challenge^i * ProveKnowledge_i(values_i) -- with the MSM over BasisExpSigma
 *                           riding beside the proof's five MSMs.  commitments[i].values (host memory, the private committed values
 *                           the hint received) and pok_out must stay valid until mi_prover_wait; challenge is copied.  The key pk
 *                           must have been loaded with the committed wires removed from K (mi_pk_desc.committed_wires).
 * Proof.WriteTo of the result: mi_proof_write(out, commitments, n, pok) = 164 + 32 n bytes (196 for the WHIR circuit). */
typedef struct mi_bsb22_input { mi_pedersen_pk *key; const mi_fr *values; size_t n; } mi_bsb22_input;
int32_t mi_prover_commit(mi_prover *p, mi_pedersen_pk *key, const mi_fr *values, size_t n, mi_g1_affine *commitment);
int32_t mi_prover_submit_bsb22(mi_prover *p, mi_pk *pk, const mi_fr *W, size_t n_wires,
                               const mi_fr *a, const mi_fr *b, const mi_fr *c, size_t n_constraints,
                               const mi_fr *r, const mi_fr *s, const mi_bsb22_input *commitments, uint32_t n_commitments,
                               const mi_fr *challenge, mi_proof_out *out, mi_g1_affine *pok_out, mi_stats *stats, uint64_t *ticket);

/* ---- device groups: one proof / one MSM point-sharded over several GPUs (BASELINE configs[4]).
 * The reference's single call groth16.Prove (mt.go:496) knows no devices; a Go caller that wants one proof spread over the
 * 8 MI355X of a node binds these.  pk points are static, so mi_pk_load_sharded cuts the wires
 * (and the N - 1 pairs of the Z MSM) into `world` contiguous ranges and keeps slice r of pk.G1.{A,B,K,Z} / pk.G2.B resident
 * on rank r; per proof only scalars move (W slices from the host, h slices device to device from the lead rank, which runs
 * computeH: NTT = replicas only).  EC addition is not an RCCL reduce op, so the exchange is byte-typed:
 *   mode 0  every rank finishes Pippenger locally; one partial sum per MSM is combined (host additions in one process,
 *           ncclAllGather(ncclUint8) with one rank per process);
 *   mode 1  "all-reduce of partial bucket sums": every rank stops at its bucket sums, rank r receives the keys it owns from
 *           every other rank (reduce-scatter as grouped ncclSend / ncclRecv, one hop on the xGMI mesh), adds them, reduces
 *           its slice; the per-rank results are combined as in mode 0.
 * Results are bit-identical to the unsharded entry points.  A group serves ONE call at a time: an entry point called while another
 * call on the same group is still running returns MI_EINVAL at once and touches nothing (the exchange streams and receive
 * buffers belong to the running call).  A process holds either ALL ranks of a group (mi_group_create) or exactly ONE
 * (mi_group_create_rank); with one rank per process every process makes the same calls in the same order (they are
 * collectives), each with its own rank's data.
 * Failures are collective too: before every exchange the ranks agree on their status (one small all-gather), so a call either succeeds
 * on every rank or returns an error on every rank -- the failing rank its own, the others "rank r failed" -- with every MSM slot
 * drained; nobody is left waiting in an exchange.  Only a failure INSIDE an exchange (a dead peer, an RCCL error) breaks the group:
 * later calls on it return MI_EHIP until it is destroyed and created anew. ---- */
typedef struct mi_group mi_group;
typedef struct mi_pk_sharded mi_pk_sharded;
/* all ranks in this process, one context per entry of dev_ids (SURVEY 8b proposed mi_init(dev_ids, n_dev, ...)).  Distinct
 * devices: RCCL communicator (ncclCommInitAll).  A device named twice (1-GPU rehearsal): same-process peer copies. */
int32_t mi_group_create(const int *dev_ids, int n_dev, mi_group **out);
/* one rank per process: id = mi_group_unique_id() from rank 0, handed to the others by the caller's own channel */
int32_t mi_group_unique_id(uint8_t id[128]);
int32_t mi_group_create_rank(int device_id, int rank, int world, const uint8_t id[128], mi_group **out);   /* = _ex(..., MI_GROUP_TRANSPORT_RCCL, ...) */
/* the same with the transport named.  MI_GROUP_TRANSPORT_HOST: the processes meet in a POSIX shared-memory segment named after the 128
 * id bytes (any 128 bytes all ranks share; mi_group_unique_id is not needed) and slices travel device -> segment -> device.  For ranks
 * RCCL cannot connect: two processes on ONE device (RCCL refuses two ranks per device; how a 1-GPU box rehearses this flow) or a box
 * without a working RCCL fabric.
 * THE DEAD-PEER CONTRACT, both transports: with one rank per process no call of this library waits for another rank without a deadline.
 * MI_GROUP_TIMEOUT_MS (environment, read when the group is created; default 60000) bounds the time a rank waits for its peers while
 * NOTHING completes -- joining the group, an exchange, an all-gather.  When it passes (a peer's process ended, a link went down), or
 * when RCCL reports an asynchronous error, the call returns MI_EHIP with a message that says "timeout", the group is broken (every
 * later call on it returns MI_EHIP at once) and must be destroyed; mi_group_destroy itself does not wait for anybody.
 *   RCCL transport: the per-rank communicator is non-blocking (ncclCommInitRankConfig, blocking = 0); joining, every group of sends /
 *     receives and every all-gather is polled with ncclCommGetAsyncError / hipStreamQuery against the deadline, and ncclCommAbort takes
 *     the communicator's kernels off the stream when it passes.  An exchange has completed on every rank that returns from it.
 *   host-staged transport: every wait on the shared segment has the deadline, and a rank that gives up poisons the segment so that the
 *     others stop waiting at once.  MI_GROUP_SHM_CHUNK_KB (default 1024, 4..65536): bytes per ring slot of the segment.
 * (Single-process groups, mi_group_create, have no peers in other processes: their communicators stay blocking.) */
#define MI_GROUP_TRANSPORT_RCCL 1
#define MI_GROUP_TRANSPORT_HOST 3
int32_t mi_group_create_rank_ex(int device_id, int rank, int world, const uint8_t id[128], int transport, mi_group **out);
int32_t mi_group_destroy(mi_group *g);
/* The lead's share of the WIRES of a sharded key.  Rank 0 also runs computeH (the NTT does not shard: SURVEY 8e) and no rank can start
 * its Z MSM before h exists, so a lead that carries an equal share of the wire MSMs lengthens the critical path of the proof.
 * permille = the fraction of an even share (nb_wires / world) that rank 0 takes, 0..1000; the other ranks split the rest evenly; the
 * N - 1 pairs of the Z MSM are always cut evenly.  1000 = the even cut.  MI_LEAD_SHARE_AUTO (the default): 1000 for one rank, 500 for
 * two, 0 from three ranks on -- from the measured ratio computeH : wire MSMs = 1 : 2 at N = 2^26.  Set it -- to the same
 * value in every process -- BEFORE mi_pk_load_sharded*: the key's parts are cut by it (a disagreement fails that load on every rank),
 * and a caller that passes device slices (mi_pk_load_sharded_dev, mi_groth16_prove_sharded_dev) cuts its arrays by
 * mi_group_wire_range.  Same proofs whatever the share. */
#define MI_LEAD_SHARE_AUTO 0xffffffffu
int32_t mi_group_set_lead_share(mi_group *g, uint32_t permille);
/* wires [*lo, *hi) of global rank `rank` under the group's current lead share */
int32_t mi_group_wire_range(const mi_group *g, uint64_t nb_wires, int rank, uint64_t *lo, uint64_t *hi);
/* computeH OVER the ranks (2, 4, 8 or 16 of them, N >= ranks^2): every transform becomes a local size-N/ranks transform and one
 * cross-rank step between two all-to-alls over the group's transport (9 batches per computeH, each moving (ranks - 1) / ranks of a
 * slice per rank), and the h slices are born on the ranks whose Z pairs they multiply -- instead of rank 0 transforming alone while the
 * others wait for h (the cap of a proof sharded over 8 GPUs moves from ~3x to the MSMs' own 1 / ranks).  Same h, same
 * proof bytes.  on = 1: mi_groth16_prove_sharded (host arrays) then needs a and b (and c, or NULL) in EVERY process, not on rank 0's
 * alone, and takes its rank's rows from them; mi_groth16_prove_sharded_dev is unchanged (its a, b, c live on rank 0's device).
 * The same value in every process.  Default 0. */
int32_t mi_group_set_sharded_compute_h(mi_group *g, uint32_t on);
/* computeH alone, as a collective: local rank i passes device pointers to ITS rows of a, b (and c; c_sl == NULL: c = a o b on the
 * device) -- rows [r M, min((r + 1) M, n_constraints)) of global rank r, M = N / ranks -- and receives its M coefficients of h in
 * gnark's bit-reversed order (rank r: positions [r M, (r + 1) M) of what mi_compute_h_dev returns). */
int32_t mi_compute_h_sharded_dev(mi_group *g, uint32_t log_n, const mi_fr *const *a_sl, const mi_fr *const *b_sl, const mi_fr *const *c_sl,
                                 size_t n_constraints, mi_fr *const *h_sl);
int32_t mi_group_world(const mi_group *g);
int32_t mi_group_local(const mi_group *g);                 /* ranks held by this process */
mi_ctx *mi_group_ctx(mi_group *g, int local_rank);         /* for mi_dev_* / generators on that rank's device */
const char *mi_group_last_error(mi_group *g);
int32_t mi_group_rank(const mi_group *g);                  /* global rank of this process's first local rank */
int32_t mi_group_transport(const mi_group *g);             /* 1 = RCCL, 2 = copies inside one process, 3 = host-staged (shared memory) */
/* transport check: every rank sends `bytes` patterned bytes to every rank (itself included) and verifies what it received */
int32_t mi_group_exchange_selftest(mi_group *g, size_t bytes);
/* desc: the same whole-key descriptor as mi_pk_load (host arrays); with one rank per process every process passes the whole
 * descriptor and keeps its own rank's slice.  The fixed-base table plan is agreed over the whole group (tightest device). */
int32_t mi_pk_load_sharded(mi_group *g, const mi_pk_desc *desc, mi_pk_sharded **out);
/* slice_descs[i], i < mi_group_local(): header (log_n, nb_public, nb_wires) and masks of the WHOLE key in host memory; the five
 * point arrays are DEVICE pointers on local rank i's device to that rank's slices (the points of wires [nb_wires r / world,
 * nb_wires (r+1) / world) and the Z pairs [(N-1) r / world, (N-1)(r+1) / world), r = global rank), counts = points of the slice.
 * Adopted by reference like mi_pk_load_dev (the caller keeps them alive). */
int32_t mi_pk_load_sharded_dev(mi_group *g, const mi_pk_desc *slice_descs, mi_pk_sharded **out);
int32_t mi_pk_sharded_free(mi_group *g, mi_pk_sharded *pk);
/* groth16.Prove after the solve (mt.go:496) over the ranks of the group; arguments as mi_groth16_prove.  W is the WHOLE wire
 * vector: a process reads only the wire ranges of its local ranks.  a, b, c are read by the process that holds rank 0 (which runs
 * computeH and hands every rank its slice of h over the group's transport); other processes may pass NULL.  Every process
 * receives the proof. */
int32_t mi_groth16_prove_sharded(mi_group *g, mi_pk_sharded *pk, const mi_fr *W, size_t n_wires,
                                 const mi_fr *a, const mi_fr *b, const mi_fr *c, size_t n_constraints,
                                 const mi_fr *r, const mi_fr *s, uint32_t mode, mi_proof_out *out, mi_stats *stats_or_null);
/* the same with the inputs already in HBM: W_dev[i] = the wire range of local rank i on its device (n_wires = the WHOLE count);
 * a_dev, b_dev, c_dev on rank 0's device (NULL in the other processes) */
int32_t mi_groth16_prove_sharded_dev(mi_group *g, mi_pk_sharded *pk, const mi_fr *const *W_dev, size_t n_wires,
                                     const mi_fr *a_dev, const mi_fr *b_dev, const mi_fr *c_dev, size_t n_constraints,
                                     const mi_fr *r, const mi_fr *s, uint32_t mode, mi_proof_out *out, mi_stats *stats_or_null);
/* the same with a, b, c as ROW SLICES per local rank (rows [r M, min((r + 1) M, n_constraints)) of global rank r on that rank's device,
 * M = N / ranks; c_sl == NULL: c = a o b): computeH runs over the ranks (mi_group_set_sharded_compute_h says what that means; here it
 * is the only way, whatever the group's setting).  2, 4, 8 or 16 ranks, N >= ranks^2. */
int32_t mi_groth16_prove_sharded_slices_dev(mi_group *g, mi_pk_sharded *pk, const mi_fr *const *W_dev, size_t n_wires,
                                            const mi_fr *const *a_sl, const mi_fr *const *b_sl, const mi_fr *const *c_sl, size_t n_constraints,
                                            const mi_fr *r, const mi_fr *s, uint32_t mode, mi_proof_out *out, mi_stats *stats);
/* one MSM over host arrays cut into contiguous slices (single-process groups) */
int32_t mi_msm_g1_sharded(mi_group *g, const mi_g1_affine *pts, const mi_fr *scalars, size_t n, uint32_t flags,
                          uint32_t mode, mi_g1_jac *out);
int32_t mi_msm_g2_sharded(mi_group *g, const mi_g2_affine *pts, const mi_fr *scalars, size_t n, uint32_t flags,
                          uint32_t mode, mi_g2_jac *out);
/* one MSM whose pairs already sit on the ranks' devices: arrays of mi_group_local() device pointers / counts; n_total = pairs
 * over ALL ranks (every rank passes the same value: it fixes the common window width).  Every rank receives the result. */
int32_t mi_msm_g1_sharded_dev(mi_group *g, const mi_g1_affine *const *pts_dev, const mi_fr *const *scalars_dev,
                              const size_t *n_local, size_t n_total, uint32_t flags, uint32_t mode, mi_g1_jac *out);
int32_t mi_msm_g2_sharded_dev(mi_group *g, const mi_g2_affine *const *pts_dev, const mi_fr *const *scalars_dev,
                              const size_t *n_local, size_t n_total, uint32_t flags, uint32_t mode, mi_g2_jac *out);

/* ---- Proof.WriteTo / point encoding (row a12), pure host code ---- */
void mi_g1_compress(const mi_g1_affine *p, uint8_t out[32]);
void mi_g2_compress(const mi_g2_affine *p, uint8_t out[64]);
/* Ar | Bs | Krs | u32-BE n_commitments | commitments | commitment_pok ; returns bytes written
 * (164 + 32*n_commitments).  commitment_pok may be NULL (encoded as infinity). */
size_t mi_proof_write(const mi_proof_out *proof, const mi_g1_affine *commitments,
                      uint32_t n_commitments, const mi_g1_affine *commitment_pok, uint8_t *out);

/* ---- BSB22 Pedersen commitment key: replaces gnark-crypto ecc/bn254/fr/pedersen ProvingKey.Commit /
 * ProveKnowledge / Fold, which gnark's prover calls because the WHIR circuit uses lookups
 * (/root/reference/utilities/utilities.go:189 logderivlookup.New; mtUtilities.go:452 uints.New).  Commit runs INSIDE the
 * solve (hint override), so it is a synchronous low-latency call; the SHA-256 hash-to-field of the commitment stays in Go.
 *   Commit(values)          = sum_i values[i] * Basis[i]
 *   ProveKnowledge(values)  = sum_i values[i] * BasisExpSigma[i]
 *   Fold(points, challenge) = sum_i challenge^i * points[i]          (pedersen.Fold / FoldCommitments)            ---- */
int32_t mi_pedersen_pk_load(mi_ctx *ctx, const mi_g1_affine *basis, const mi_g1_affine *basis_exp_sigma, size_t n,
                            mi_pedersen_pk **out);                       /* uploads once, device-resident */
int32_t mi_pedersen_pk_free(mi_ctx *ctx, mi_pedersen_pk *pk);
int32_t mi_pedersen_commit(mi_ctx *ctx, mi_pedersen_pk *pk, const mi_fr *values, size_t n, mi_g1_affine *commitment);
int32_t mi_pedersen_prove_knowledge(mi_ctx *ctx, mi_pedersen_pk *pk, const mi_fr *values, size_t n, mi_g1_affine *pok);
int32_t mi_pedersen_fold(const mi_g1_affine *points, size_t n, const mi_fr *challenge, mi_g1_affine *out); /* host */

/* ---- fixed-base batch scalar multiplication: replaces gnark-crypto ecc/bn254
 * BatchScalarMultiplicationG1 / G2, the bulk of groth16.Setup (/root/reference/mt.go:448: every pk / vk point is
 * scalar * generator).  out[i] = scalars[i] * base, affine.  Windowed table of the base built on the device, one
 * thread per scalar. ---- */
int32_t mi_batch_scalar_mul_g1(mi_ctx *ctx, const mi_g1_affine *base, const mi_fr *scalars, size_t n, mi_g1_affine *out);
int32_t mi_batch_scalar_mul_g1_dev(mi_ctx *ctx, const mi_g1_affine *base, const mi_fr *scalars_dev, size_t n, mi_g1_affine *out_dev);
int32_t mi_batch_scalar_mul_g2(mi_ctx *ctx, const mi_g2_affine *base, const mi_fr *scalars, size_t n, mi_g2_affine *out);
int32_t mi_batch_scalar_mul_g2_dev(mi_ctx *ctx, const mi_g2_affine *base, const mi_fr *scalars_dev, size_t n, mi_g2_affine *out_dev);

/* ---- ProveKit artefact ingestion: what /root/reference/main.go:92-152 and the top of verify_circuit (mt.go:306-401) do
 * with the files the Rust prover wrote, BEFORE frontend.Compile -- pure host code, no device needed.  Their consumers are gnark's
 * frontend and solver (Go), so a Go caller binds these only to replace go-ark-serialize + the decoding loops; nothing of the prove path
 * depends on them.  The arkworks wire format is restated from the published ark-serialize rules (go-ark-serialize, go.mod:10, is absent
 * from the reference tree): parity unpinned until a real ProveKit `proof` file is decoded.
 *   mi_whir_proof_decode        go_ark_serialize.CanonicalDeserializeWithMode(proofFile, &proof, false, false), main.go:101, into
 *                               ProofObject (main.go:35-39: round0_merkle_paths, merkle_paths, statement_values_at_random_point)
 *   mi_whir_element_shape       leaves proved, tree height (= len(AuthPathsSuffixes[0]), mt.go:243), leaf values of one ProofElement
 *   mi_whir_parse_paths         ParsePathsObject, mt.go:229-304, for one ProofElement: auth_paths[j][z] = node z (leaf end first) of
 *                               leaf j's authentication path after PrefixDecodePath + Reverse; leaves reduced mod r (LimbsToBigIntMod)
 *   mi_whir_reverse             utilities.Reverse, utilities/utilities.go:58-65 (out must not alias in)
 *   mi_whir_prefix_decode_path  utilities.PrefixDecodePath, utilities/utilities.go:67-78 (MI_EINVAL where Go would panic: prefix_len > n_prev)
 *   mi_whir_limbs_to_fr         typeConverters.LimbsToBigIntMod, typeConverters/typeConverters.go:26-44: 4 x u64 little-endian limbs
 *                               -> the canonical value mod r, same limb order (NOT Montgomery: multiply by R for an mi_fr)
 *   mi_whir_interner_decode     Interner{Values []Fp256}, main.go:74-76,146
 *   mi_whir_matrix_cells        the CSR -> MatrixCell loops of mt.go:358-401 (row i owns entries [row_indices[i], row_indices[i+1] - 1],
 *                               the last row runs to the end; value = LimbsToBigIntMod(interner[values[j]]))
 *   mi_whir_config_parse        json.Unmarshal into Config, main.go:41-58,115: unknown keys ignored, missing keys zero, null leaves a field as
 *                               it is (a top-level null too), keys match ASCII-case-insensitively, the last duplicate wins; strict literals,
 *                               numbers (no leading zeros; an int refuses fractions, exponents, values outside int64) and string escapes
 *                               (an unpaired \uD800-\uDFFF escape becomes U+FFFD); nothing but white space may follow the object; nesting
 *                               deeper than 10000 is refused.  `transcript` as a JSON array of numbers or a padded base64 string (\r, \n
 *                               skipped); decimal strings -> 4 x u64 limbs.  Refused although Go would accept: more than
 *                               MI_WHIR_MAX_ROUNDS list entries, a decimal string that is empty / not a number / >= 2^256 (Go keeps the string
 *                               and fails later, mt.go:310,352).  Copied as they are although Go substitutes U+FFFD: invalid UTF-8 bytes
 *                               inside a string.
 * These readers take bytes an outside party wrote: they run under AddressSanitizer / UBSan with a mutation driver on every CPU test run
 * (gnark-whir_amd/Makefile `sanitize`, tests/test_parsers_sanitized.py), together with mi_pk_raw_inspect. ---- */
typedef struct mi_whir_proof mi_whir_proof;
typedef struct mi_whir_shape { uint64_t n_leaves, tree_height, total_leaf_values; } mi_whir_shape;
int32_t mi_whir_proof_decode(const uint8_t *buf, size_t len, mi_whir_proof **out, size_t *consumed_or_null);
void mi_whir_proof_free(mi_whir_proof *p);
uint64_t mi_whir_proof_elements(const mi_whir_proof *p, int which /* 0 = round0_merkle_paths, 1 = merkle_paths */);
uint64_t mi_whir_proof_statement_values(const mi_whir_proof *p, uint64_t *limbs_out /* count x 4 raw limbs, may be NULL */);
int32_t mi_whir_element_shape(const mi_whir_proof *p, int which, uint64_t i, mi_whir_shape *out);
int32_t mi_whir_parse_paths(const mi_whir_proof *p, int which, uint64_t i, uint8_t *auth_paths /* n_leaves x tree_height x 32 */,
                            uint8_t *leaf_sibling_hashes /* n_leaves x 32 */, uint64_t *leaf_indexes /* n_leaves */,
                            uint64_t *leaf_lengths /* n_leaves */, uint64_t *leaves /* total_leaf_values x 4 */);   /* any output may be NULL */
int32_t mi_whir_reverse(const void *in, size_t n, size_t elem_bytes, void *out);
int32_t mi_whir_prefix_decode_path(const void *prev, size_t n_prev, uint64_t prefix_len, const void *suffix, size_t n_suffix,
                                   size_t elem_bytes, void *out /* (prefix_len + n_suffix) elements */, size_t *n_out);
void mi_whir_limbs_to_fr(const uint64_t limbs[4], uint64_t out[4]);
int32_t mi_whir_interner_decode(const uint8_t *buf, size_t len, uint64_t *limbs_out /* count x 4, may be NULL */, uint64_t *n_out, size_t *consumed_or_null);
int32_t mi_whir_matrix_cells(const uint64_t *row_indices, size_t n_rows, const uint64_t *col_indices, const uint64_t *values, size_t nnz,
                             const uint64_t *interner_limbs, size_t n_interner, uint64_t *rows_out, uint64_t *cols_out, uint64_t *values_out /* nnz x 4 */);
#define MI_WHIR_MAX_ROUNDS 64
typedef struct mi_whir_config {   /* Config, main.go:41-58; pointers are owned by the config (mi_whir_config_free) */
    int64_t log_num_constraints, n_rounds, n_vars, final_queries, final_pow_bits, final_folding_pow_bits, rate, transcript_len;
    int64_t folding_factor[MI_WHIR_MAX_ROUNDS], ood_samples[MI_WHIR_MAX_ROUNDS], num_queries[MI_WHIR_MAX_ROUNDS], pow_bits[MI_WHIR_MAX_ROUNDS];
    uint32_t n_folding_factor, n_ood_samples, n_num_queries, n_pow_bits;
    uint64_t domain_generator[4];              /* the decimal string as an integer (mt.go:310), little-endian limbs */
    const char *io_pattern; size_t io_pattern_len;
    const uint8_t *transcript; size_t n_transcript;
    const uint64_t *statement_evaluations; size_t n_statement_evaluations;   /* decimal strings (mt.go:352) -> 4 limbs each */
    void *store;
} mi_whir_config;
int32_t mi_whir_config_parse(const char *json, size_t len, mi_whir_config **out);
void mi_whir_config_free(mi_whir_config *c);

/* ---- partial-sum combine for the point-sharded MSM (SURVEY section 8e option i): adds n
 * Jacobian partial results (e.g. all-gathered from the ranks), host side ---- */
int32_t mi_g1_sum(const mi_g1_jac *parts, size_t n, mi_g1_jac *out);
int32_t mi_g2_sum(const mi_g2_jac *parts, size_t n, mi_g2_jac *out);

/* ---- device-side test / bench utilities (not part of the reference surface) ---- */
#define MI_DIST_UNIFORM 0
#define MI_DIST_WHIR 1      /* 45% {0,1}, 25% bytes, 5% 64-bit, 25% uniform */
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
 *   "finisher_max" 0..2^20   0 = automatic (G1 4096, G2 1024)
 *   "finisher_min_level" 0..16   the finisher follows accumulate pass number this + 1 at the earliest (default 2: the first two passes
 *                            are where every ordinary bucket ends; a finisher over 2^19 buckets of 13 partial sums each measured -5 %)
 *   "item_l1", "item_l2", "reduce_seg"   = the L1, L2, seg of mi_debug_set_msm_plan, one at a time (a flat sort keeps its own item size: "flat_item_l1")
 *   "hold_accum" 0 | 1       = mi_debug_set_prove_schedule
 *   "ntt_lds_floor_kb" 0..160   LDS every NTT pass workgroup requests at least (caps the workgroups per CU) */
int32_t mi_debug_set_knob(mi_ctx *ctx, const char *name, int64_t value);
/* Counters the tests read to prove that an optional path really ran: "z_count_fused_launches" = computeH last launches of this context that
 * carried the Z MSM's digit count (knob "z_count_fused").  MI_EINVAL for an unknown name. */
int32_t mi_debug_get_counter(mi_ctx *ctx, const char *name, uint64_t *out);
/* Process-wide, for contexts created afterwards: how the MSM slots of a context share streams (0: K's stream created, destroyed and
 * pointed at B1's, as rounds 4-5 did; 1, the default: never created; 2: A, B1 and K on one stream).  Same results; an experiment on which
 * chains end up on one hardware queue. */
int32_t mi_debug_set_stream_plan(int32_t plan);
/* Tracing: on != 0 wraps the host-side phases of every call -- uploads, computeH's enqueue, each MSM's enqueue and
 * collection, the pool's stages, the device groups' exchanges -- in roctx ranges ("mi.prove", "mi.computeH.enqueue", "mi.msm.Z.enqueue", ...),
 * which `rocprofv3 --marker-trace --kernel-trace` shows beside the kernels.  Process-wide; off by default (one relaxed load per site).
 * MI_ENODEV when no roctx library can be loaded (librocprofiler-sdk-roctx / libroctx64, looked up at the first call; nothing links it). */
int32_t mi_debug_set_trace_ranges(int32_t on);
/* error-path tests: the nth MI-checked HIP call from now (library-wide, any thread) fails with hipErrorUnknown instead of
 * running; 0 disarms.  Used to prove that init / load / prove unwind without leaks or crashes. */
int32_t mi_debug_inject_hip_failure(int32_t nth);
/* raw device memory helpers so hosts without a HIP binding (ctypes, cgo) can stage data */
int32_t mi_dev_alloc(mi_ctx *ctx, size_t bytes, void **out_dev);
int32_t mi_dev_free(mi_ctx *ctx, void *dev);
int32_t mi_dev_upload(mi_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);
int32_t mi_dev_download(mi_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);
int32_t mi_dev_sync(mi_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif
