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
 *
 * Companion headers (same library, same conventions; each includes this one):
 *   mi355x_groth16_group.h   one proof / one MSM point-sharded over several GPUs
 *   mi355x_whir_ingest.h     host-only decoding of ProveKit's artefacts
 *   mi355x_groth16_debug.h   generators, probes, plan knobs, fault injection: tests, bench and tuning ONLY -- a service that binds
 *                            the prove path never includes it
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

/* ---- tracing: on != 0 wraps the host-side phases of every call -- uploads, computeH's enqueue, each MSM's enqueue and
 * collection, the pool's stages, the device groups' exchanges -- in roctx ranges ("mi.prove", "mi.computeH.enqueue", "mi.msm.Z.enqueue", ...),
 * which `rocprofv3 --marker-trace --kernel-trace` shows beside the kernels.  Process-wide; off by default (one relaxed load per site).
 * MI_ENODEV when no roctx library can be loaded (looked up at the first call; nothing links it). */
int32_t mi_set_trace_ranges(int32_t on);

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