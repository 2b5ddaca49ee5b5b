// C++ host-side mirror of the reference's operator surface for the hot path, above the C-ABI of
// include/mi355x_groth16.h.  The reference is Go (gnark); no Go toolchain exists in the build image, so the
// compiled-language host side is written in C++ with gnark's names and argument meaning:
//
//   gnark (mt.go:447-497)                                   here
//   ------------------------------------------------------  ------------------------------------------------
//   pk, vk, _ := groth16.Setup(ccs)                         groth16::ProvingKey pk(ctx, desc)   // device-resident
//   proof, err := groth16.Prove(ccs, pk, witness, opts...)  groth16::Proof proof = groth16::Prove(ctx, pk, solution, r, s)
//   proof.WriteTo(w)                                        proof.WriteTo(bytes)
//
// `solution` is what gnark's r1cs.Solve hands to the prover (W, A, B, C); r and s are the two blinding scalars
// gnark samples with fr.SetRandom (inputs here so that CPU and GPU proofs of the same data are byte-identical).
// Errors surface as groth16::Error (gnark returns `error`); nothing is swallowed.
#pragma once
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>
#include "../../include/mi355x_groth16.h"
#include "../../include/mi355x_groth16_group.h"
#include "../../include/mi355x_groth16_debug.h"   // (this mirror is the TEST side: generators and knobs)

namespace groth16 {

struct Error : std::runtime_error {
    int32_t code;
    Error(int32_t c, const std::string &what) : std::runtime_error(what), code(c) {}
};

class Context {
  public:
    explicit Context(int device = 0) {
        int32_t rc = mi_init(device, &ctx_);
        if (rc != MI_OK) throw Error(rc, "mi_init failed (no gfx950 device? there is no CPU path)");
    }
    ~Context() { if (ctx_) mi_shutdown(ctx_); }
    Context(const Context &) = delete;
    Context &operator=(const Context &) = delete;
    mi_ctx *get() const { return ctx_; }
    void check(int32_t rc) const { if (rc != MI_OK) throw Error(rc, mi_last_error(ctx_)); }

  private:
    mi_ctx *ctx_ = nullptr;
};

// Solution vectors of constraint/bn254 R1CSSolution: W (all wires), A, B, C (one value per constraint).
struct Solution {
    const mi_fr *W; size_t nWires;
    const mi_fr *A, *B, *C; size_t nConstraints;
};

// groth16/bn254 Proof
struct Proof {
    mi_g1_affine Ar;
    mi_g2_affine Bs;
    mi_g1_affine Krs;
    std::vector<mi_g1_affine> Commitments;   // BSB22 (filled by the caller when the circuit commits)
    mi_g1_affine CommitmentPok{};            // all-zero = point at infinity
    // Proof.WriteTo: Ar | Bs | Krs | u32-BE len | Commitments | CommitmentPok, compressed points
    void WriteTo(std::vector<uint8_t> &out) const {
        mi_proof_out p{Ar, Bs, Krs};
        out.resize(164 + 32 * Commitments.size());
        size_t n = mi_proof_write(&p, Commitments.data(), (uint32_t)Commitments.size(), &CommitmentPok, out.data());
        out.resize(n);
    }
};

// Device-resident proving key (the counterpart of gnark's icicle ProvingKey with its G1Device/G2Device members).
class ProvingKey {
  public:
    ProvingKey(const Context &ctx, const mi_pk_desc &desc) : ctx_(ctx) { ctx_.check(mi_pk_load(ctx_.get(), &desc, &pk_)); }
    ~ProvingKey() { if (pk_) mi_pk_free(ctx_.get(), pk_); }
    ProvingKey(const ProvingKey &) = delete;
    ProvingKey &operator=(const ProvingKey &) = delete;
    mi_pk *get() const { return pk_; }

  private:
    const Context &ctx_;
    mi_pk *pk_ = nullptr;
};

// groth16.Prove after the solve.
inline Proof Prove(const Context &ctx, const ProvingKey &pk, const Solution &s, const mi_fr &r, const mi_fr &sBlind, mi_stats *stats = nullptr) {
    mi_proof_out out{};
    ctx.check(mi_groth16_prove(ctx.get(), pk.get(), s.W, s.nWires, s.A, s.B, s.C, s.nConstraints, &r, &sBlind, &out, stats));
    Proof p;
    p.Ar = out.ar; p.Bs = out.bs; p.Krs = out.krs;
    return p;
}

// Several groth16.Prove calls in flight on one GPU (what goroutines calling Prove concurrently get from gnark's CPU
// prover): Submit queues a proof and returns a handle, Wait blocks for it.  One device-resident key serves every context.
class Prover {
  public:
    class Pending {
      public:
        Proof Wait() {   // once
            int32_t rc = mi_prover_wait(p_, ticket_);
            if (rc != MI_OK) throw Error(rc, mi_prover_last_error(p_));
            Proof pr;
            pr.Ar = out_->ar; pr.Bs = out_->bs; pr.Krs = out_->krs;
            return pr;
        }
      private:
        friend class Prover;
        mi_prover *p_ = nullptr;
        uint64_t ticket_ = 0;
        std::unique_ptr<mi_proof_out> out_;   // written by a worker thread until Wait returns
    };
    explicit Prover(int device = 0, uint32_t inFlight = 3) {
        int32_t rc = mi_prover_create(device, inFlight, &p_);
        if (rc != MI_OK) throw Error(rc, "mi_prover_create failed (no gfx950 device? there is no CPU path)");
    }
    ~Prover() { if (p_) mi_prover_destroy(p_); }
    Prover(const Prover &) = delete;
    Prover &operator=(const Prover &) = delete;
    mi_ctx *ctx(uint32_t i = 0) const { return mi_prover_ctx(p_, i); }   // for mi_pk_load
    // the solution vectors must stay alive until Wait returns
    Pending Submit(mi_pk *pk, const Solution &s, const mi_fr &r, const mi_fr &sBlind) {
        Pending h;
        h.p_ = p_;
        h.out_.reset(new mi_proof_out{});
        int32_t rc = mi_prover_submit(p_, pk, s.W, s.nWires, s.A, s.B, s.C, s.nConstraints, &r, &sBlind, h.out_.get(), nullptr, &h.ticket_);
        if (rc != MI_OK) throw Error(rc, "mi_prover_submit failed");
        return h;
    }

  private:
    mi_prover *p_ = nullptr;
};

// One groth16.Prove spread over several GPUs of a node: contexts on `devices`, the key point-sharded at load,
// mode 0 = per-device partial sums, 1 = reduce-scatter of bucket sums (RCCL; include/mi355x_groth16.h, device groups).
class DeviceGroup {
  public:
    explicit DeviceGroup(const std::vector<int> &devices) {
        int32_t rc = mi_group_create(devices.data(), (int)devices.size(), &g_);
        if (rc != MI_OK) throw Error(rc, "mi_group_create failed (no gfx950 device? there is no CPU path)");
    }
    ~DeviceGroup() { if (spk_) mi_pk_sharded_free(g_, spk_); if (g_) mi_group_destroy(g_); }
    DeviceGroup(const DeviceGroup &) = delete;
    DeviceGroup &operator=(const DeviceGroup &) = delete;
    void check(int32_t rc) const { if (rc != MI_OK) throw Error(rc, mi_group_last_error(g_)); }
    void LoadKey(const mi_pk_desc &desc) { if (spk_) { mi_pk_sharded_free(g_, spk_); spk_ = nullptr; } check(mi_pk_load_sharded(g_, &desc, &spk_)); }
    Proof Prove(const Solution &s, const mi_fr &r, const mi_fr &sBlind, uint32_t mode = 0) {
        mi_proof_out out{};
        check(mi_groth16_prove_sharded(g_, spk_, s.W, s.nWires, s.A, s.B, s.C, s.nConstraints, &r, &sBlind, mode, &out, nullptr));
        Proof p;
        p.Ar = out.ar; p.Bs = out.bs; p.Krs = out.krs;
        return p;
    }
    mi_g1_jac MultiExpG1(const std::vector<mi_g1_affine> &points, const std::vector<mi_fr> &scalars, uint32_t mode = 0) {
        if (points.size() != scalars.size()) throw Error(MI_EINVAL, "MultiExp: len(points) != len(scalars)");
        mi_g1_jac out{};
        check(mi_msm_g1_sharded(g_, points.data(), scalars.data(), points.size(), 0, mode, &out));
        return out;
    }

  private:
    mi_group *g_ = nullptr;
    mi_pk_sharded *spk_ = nullptr;
};

// ecc/bn254 MultiExp
inline mi_g1_jac MultiExpG1(const Context &ctx, const std::vector<mi_g1_affine> &points, const std::vector<mi_fr> &scalars) {
    if (points.size() != scalars.size()) throw Error(MI_EINVAL, "MultiExp: len(points) != len(scalars)");   // gnark: same error
    mi_g1_jac out{};
    ctx.check(mi_msm_g1(ctx.get(), points.data(), scalars.data(), points.size(), 0, &out));
    return out;
}
inline mi_g2_jac MultiExpG2(const Context &ctx, const std::vector<mi_g2_affine> &points, const std::vector<mi_fr> &scalars) {
    if (points.size() != scalars.size()) throw Error(MI_EINVAL, "MultiExp: len(points) != len(scalars)");
    mi_g2_jac out{};
    ctx.check(mi_msm_g2(ctx.get(), points.data(), scalars.data(), points.size(), 0, &out));
    return out;
}
// fft.Domain: FFT / FFTInverse with fft.DIF / fft.DIT and fft.OnCoset()
enum Decimation { DIF = 0, DIT = 1 };
inline void FFT(const Context &ctx, std::vector<mi_fr> &a, uint32_t logN, Decimation d, bool onCoset = false) {
    if (a.size() != ((size_t)1 << logN)) throw Error(MI_EINVAL, "FFT: len(a) != domain cardinality");
    ctx.check(mi_ntt(ctx.get(), a.data(), logN, (d == DIT ? MI_NTT_DIT : 0u) | (onCoset ? MI_NTT_COSET : 0u)));
}
inline void FFTInverse(const Context &ctx, std::vector<mi_fr> &a, uint32_t logN, Decimation d, bool onCoset = false) {
    if (a.size() != ((size_t)1 << logN)) throw Error(MI_EINVAL, "FFTInverse: len(a) != domain cardinality");
    ctx.check(mi_ntt(ctx.get(), a.data(), logN, MI_NTT_INVERSE | (d == DIT ? MI_NTT_DIT : 0u) | (onCoset ? MI_NTT_COSET : 0u)));
}

}  // namespace groth16
