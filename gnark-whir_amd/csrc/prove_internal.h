// Internals of the prove path shared by prove.hip (one device) and group.hip (point-sharded over several devices).
#pragma once
#include "ctx.h"
#include "curve.cuh"
#include "msm_curve_ops.h"
#include <functional>
#include <future>

struct mi_pk {
    u32 log_n = 0, nb_public = 0;
    u64 nb_wires = 0;
    void *g1_a = nullptr, *g1_b = nullptr, *g1_k = nullptr, *g1_z = nullptr, *g2_b = nullptr;
    u64 n_a = 0, n_b = 0, n_k = 0, n_z = 0;
    bool owns_points = false;
    u32 *idx_a = nullptr, *idx_b = nullptr, *idx_k = nullptr;  // wire index of every A / B / K point
    // pk.G1.A and pk.G1.K re-expanded to one slot per wire (zero = infinity where the wire has no point): both are multiplied
    // by W itself, so ONE sort of W serves both MSMs and neither needs a gather (prove.hip, step 5)
    G1Aff *a_full = nullptr, *k_full = nullptr;
    // Fixed-base window copies 2^(c*w) * P (msm2_core.cuh) of the bases, per group of MSMs that share a sort: A+K, B1+B2, Z.
    // c = 0: the group runs the generic path on the plain bases.  A group with tables no longer keeps its plain copy
    // (pre[0] is the base array) unless the caller owns it.
    u32 c_ak = 0, c_b = 0, c_z = 0;
    void *pre_a = nullptr, *pre_k = nullptr, *pre_b1 = nullptr, *pre_b2 = nullptr, *pre_z = nullptr;
    G1Aff alpha1, beta1, delta1;
    G2Aff beta2, delta2;
    // A part of a point-sharded key (group.hip) covers wires [wire_lo, wire_lo + nb_wires) and the Z pairs
    // [z_lo, z_lo + n_z_msm) of the 2^log_n - 1; a whole key has wire_lo = z_lo = 0 and n_z_msm = 2^log_n - 1.
    u64 wire_lo = 0, z_lo = 0, n_z_msm = 0;
    u32 gen_c_ak = 0, gen_c_b = 0, gen_c_z = 0;
    // The G1 arrays the level-1 accumulation gathers from (tables, or the plain bases of a group without tables) hold both
    // coordinates times 2^5: the R' = 2^261 packed form of the 9 x 29-bit kernel (msm_curve_ops.h).  Arrays of the caller
    // (mi_pk_load_dev) are never rewritten: b1_copy / z_copy are the key's own converted copies of pk.G1.B / pk.G1.Z then.
    bool rprime = false;
    void *b1_copy = nullptr, *z_copy = nullptr, *b2_copy = nullptr;   // generic-path window bits the parts of a sharded key agree on (0 = from n)
};


struct ShardRange { u64 w_lo, w_hi, z_lo, z_hi; };   // wires [w_lo, w_hi), Z pairs [z_lo, z_hi)
// mi_pk_load / mi_pk_load_dev (sr == nullptr) or one part of a sharded key (host arrays only)
// sr with device_points: the arrays are this part's slices on ctx's device (mi_pk_load_sharded_dev).
// adopt (device_points only): the key takes ownership of the five arrays; *took_arrays = true once it has (then they are released by
// the key on success and by this function on failure -- the caller must not free them again).
int32_t mi_pk_load_range(mi_ctx *ctx, const mi_pk_desc *d, mi_pk **out, bool device_points, const ShardRange *sr, bool adopt = false, bool *took_arrays = nullptr);
// a Pedersen key over device arrays the key takes ownership of (mi_pk_load_raw)
extern "C" int32_t mi_pedersen_pk_adopt(mi_ctx *ctx, void *basis_dev, void *basis_exp_sigma_dev, size_t n, mi_pedersen_pk **out);
// ProveKnowledge of one BSB22 commitment in two halves on slot 5 of ctx (beside a proof's five MSMs): host values in, affine point out
int32_t mi_pedersen_pok_enqueue(mi_ctx *ctx, mi_pedersen_pk *pk, const mi_fr *values, size_t n);
int32_t mi_pedersen_pok_collect(mi_ctx *ctx, mi_g1_affine *pok_or_null);
// The wire MSMs (A, B1, B2, K on slots 0..3) over W_dev = this key's wire range, ordered after ev_w; the Z MSM (slot 4) over
// h_dev = this key's first h coefficient, ordered after ev_h.  defer_reduce: stop each MSM at its bucket sums (group.hip
// exchanges them between devices before the reduce, SURVEY 8e option ii).
int32_t mi_prove_enqueue_wire_msms(mi_ctx *ctx, mi_pk *pk, const mi_fr *W_dev, hipEvent_t ev_w, bool defer_reduce = false);
// its two halves, each with one sort and one host-side wait for that sort's count pass: B1 + B2 (slots 1, 2), A + K (slots 0, 3).
// Independent of each other (own slots, own buffers): safe to call from two threads at once.
// accum_gate (may be null): MsmSlot::accum_gate for the group's two slots -- the sorts are enqueued at once, the bucket accumulations
// behind the event it returns.
int32_t mi_prove_enqueue_b_msms(mi_ctx *ctx, mi_pk *pk, const mi_fr *W_dev, hipEvent_t ev_w, bool defer_reduce = false, const std::function<hipEvent_t()> *accum_gate = nullptr);
int32_t mi_prove_enqueue_ak_msms(mi_ctx *ctx, mi_pk *pk, const mi_fr *W_dev, hipEvent_t ev_w, bool defer_reduce = false, const std::function<hipEvent_t()> *accum_gate = nullptr);
int32_t mi_prove_enqueue_z_msm(mi_ctx *ctx, mi_pk *pk, const mi_fr *h_dev, hipEvent_t ev_h, bool defer_reduce = false);

// mi_groth16_prove_dev over inputs that are still arriving in HBM (the prover pool's upload stage): W is resident; the wire MSMs are
// enqueued at once; abc_ready(k) must block until k of a, b, c (in that order) are resident (false = their upload failed): computeH is
// enqueued one vector at a time behind them (k = 3 is never asked for when c is null).  abc_arrived: they all were when the job was
// picked up (the steady state).  Same proof bytes.
int32_t mi_groth16_prove_dev_gated(mi_ctx *ctx, mi_pk *pk, const mi_fr *W_dev, size_t n_wires, const mi_fr *a_dev, const mi_fr *b_dev, const mi_fr *c_dev,
                                   size_t n_constraints, const mi_fr *r, const mi_fr *s, mi_proof_out *out, mi_stats *stats,
                                   const std::function<bool(int)> &abc_ready, bool abc_arrived);

// Blinding and assembly of Ar, Bs, Krs from the five MSM sums, exactly as gnark's prove.go composes them (row a9); host
// code over O(1) points.  start() launches the multiples of delta on host threads while the GPU works; have_a_b1() needs
// only the A and B1 sums and starts s*Ar, r*Bs1; finish() takes the rest.
struct ProofAssembler {
    const mi_pk *pk = nullptr;
    Fr rc, sc, krc;
    std::future<G1X> f_r, f_s, f_kr, f_sar;
    G2X s_delta2;
    G1X r_delta, s_delta, kr_delta, r_bs1;
    G1Aff ar_aff, bs1_aff;
    void start(const mi_pk *pk_, const mi_fr *r_m, const mi_fr *s_m);
    void have_a_b1(const G1X &msm_a, const G1X &msm_b1);
    void finish(const G1X &msm_k, const G2X &msm_b2, const G1X &msm_z, mi_proof_out *out);
};
