// GPU parity test written against the C++ host mirror (gnark-whir_amd/host/groth16_mi355x.hpp): a synthetic
// WHIR-shaped key and witness are proved by the product (HIP) and by the oracle's C restatement; the two
// Proof.WriteTo byte strings must be identical.  Links both libraries; only this test binary may do so.
#include <cstdio>
#include <cstring>
#include <vector>
#include "../../gnark-whir_amd/host/groth16_mi355x.hpp"

extern "C" {   // oracle/groth16_ref.c
void ref_gen_scalars(mi_fr *out, size_t n, uint64_t seed, int dist);
void ref_gen_g1(mi_g1_affine *out, size_t n, uint64_t seed);
void ref_gen_g2(mi_g2_affine *out, size_t n, uint64_t seed);
int32_t ref_field_op(int field, int op, void *z, const void *x, const void *y, size_t n);
int32_t ref_groth16_prove(const mi_pk_desc *pk, const mi_fr *W, size_t n_wires, const mi_fr *a, const mi_fr *b, const mi_fr *c,
                          size_t n_constraints, const mi_fr *r, const mi_fr *s, mi_proof_out *out, mi_fr *h_out_opt);
size_t ref_proof_write(const mi_proof_out *proof, const mi_g1_affine *commitments, uint32_t n, const mi_g1_affine *pok, uint8_t *out);
int32_t ref_msm_g1(const mi_g1_affine *pts, const mi_fr *scalars, size_t n, uint32_t flags, mi_g1_jac *out);
}

int main() {
    const uint32_t logN = 12;
    const size_t N = (size_t)1 << logN, nWires = N - 7, nPublic = 33, nConstraints = N - 3;
    std::vector<uint8_t> infA(nWires), infB(nWires);
    size_t na = 0, nb = 0;
    for (size_t j = 0; j < nWires; j++) { infA[j] = (j * 2654435761u >> 7) % 10 == 0; infB[j] = (j * 40503u >> 3) % 2 == 0; na += !infA[j]; nb += !infB[j]; }
    std::vector<mi_g1_affine> A(na), B1(nb), K(nWires - nPublic), Z(N), small(3);
    std::vector<mi_g2_affine> B2(nb), small2(2);
    ref_gen_g1(A.data(), na, 1); ref_gen_g1(B1.data(), nb, 2); ref_gen_g1(K.data(), K.size(), 3); ref_gen_g1(Z.data(), N, 4);
    ref_gen_g2(B2.data(), nb, 5); ref_gen_g1(small.data(), 3, 6); ref_gen_g2(small2.data(), 2, 7);
    mi_pk_desc d{};
    d.log_n = logN; d.nb_public = (uint32_t)nPublic; d.nb_wires = nWires;
    d.g1_a = A.data(); d.n_g1_a = na; d.g1_b = B1.data(); d.n_g1_b = nb; d.g1_k = K.data(); d.n_g1_k = K.size();
    d.g1_z = Z.data(); d.n_g1_z = N; d.g2_b = B2.data(); d.n_g2_b = nb;
    d.alpha1 = small[0]; d.beta1 = small[1]; d.delta1 = small[2]; d.beta2 = small2[0]; d.delta2 = small2[1];
    d.infinity_a = infA.data(); d.infinity_b = infB.data();
    std::vector<mi_fr> W(nWires), a(nConstraints), b(nConstraints), c(nConstraints), rs(2);
    ref_gen_scalars(W.data(), nWires, 8, 1); ref_gen_scalars(a.data(), nConstraints, 9, 1); ref_gen_scalars(b.data(), nConstraints, 10, 0);
    ref_field_op(0, 2, c.data(), a.data(), b.data(), nConstraints);
    ref_gen_scalars(rs.data(), 2, 11, 0);

    try {
        groth16::Context ctx(0);
        groth16::ProvingKey pk(ctx, d);
        groth16::Solution sol{W.data(), nWires, a.data(), b.data(), c.data(), nConstraints};
        groth16::Proof proof = groth16::Prove(ctx, pk, sol, rs[0], rs[1]);
        std::vector<uint8_t> got;
        proof.WriteTo(got);
        mi_proof_out want{};
        if (ref_groth16_prove(&d, W.data(), nWires, a.data(), b.data(), c.data(), nConstraints, &rs[0], &rs[1], &want, nullptr) != 0) { std::puts("oracle failed"); return 2; }
        std::vector<uint8_t> wb(164);
        wb.resize(ref_proof_write(&want, nullptr, 0, nullptr, wb.data()));
        if (got != wb) { std::puts("MISMATCH: proof bytes differ"); return 1; }
        // MultiExp through the mirror, and gnark's error behaviour on mismatched lengths
        std::vector<mi_fr> sc(A.begin() == A.end() ? 0 : 100);
        ref_gen_scalars(sc.data(), sc.size(), 12, 0);
        std::vector<mi_g1_affine> p100(A.begin(), A.begin() + 100);
        mi_g1_jac g = groth16::MultiExpG1(ctx, p100, sc), w{};
        ref_msm_g1(p100.data(), sc.data(), 100, 0, &w);
        if (std::memcmp(&g, &w, sizeof(g)) != 0) { std::puts("MISMATCH: MultiExp"); return 1; }
        bool threw = false;
        try { sc.pop_back(); groth16::MultiExpG1(ctx, p100, sc); } catch (const groth16::Error &) { threw = true; }
        if (!threw) { std::puts("MISMATCH: length check"); return 1; }
        // a witness that does not match the key must be rejected, not proved
        threw = false;
        try { groth16::Solution bad{W.data(), nWires - 1, a.data(), b.data(), c.data(), nConstraints}; groth16::Prove(ctx, pk, bad, rs[0], rs[1]); }
        catch (const groth16::Error &) { threw = true; }
        if (!threw) { std::puts("MISMATCH: size check"); return 1; }
        // four proofs in flight on a pool of two contexts sharing the same key: same bytes, job for job
        {
            groth16::Prover pool(0, 2);
            std::vector<groth16::Prover::Pending> pending;
            for (int k = 0; k < 4; k++) pending.push_back(pool.Submit(pk.get(), sol, rs[0], rs[1]));
            for (auto &h : pending) {
                std::vector<uint8_t> pb;
                h.Wait().WriteTo(pb);
                if (pb != wb) { std::puts("MISMATCH: pool proof bytes differ"); return 1; }
            }
            threw = false;
            try { groth16::Solution bad{W.data(), nWires - 1, a.data(), b.data(), c.data(), nConstraints}; pool.Submit(pk.get(), bad, rs[0], rs[1]).Wait(); }
            catch (const groth16::Error &) { threw = true; }
            if (!threw) { std::puts("MISMATCH: pool size check"); return 1; }
        }
        // the same proof over a device group: two ranks (both on device 0 here: one GPU), both exchange modes
        {
            groth16::DeviceGroup grp({0, 0});
            grp.LoadKey(d);
            for (uint32_t mode = 0; mode < 2; mode++) {
                std::vector<uint8_t> pb;
                grp.Prove(sol, rs[0], rs[1], mode).WriteTo(pb);
                if (pb != wb) { std::printf("MISMATCH: sharded proof bytes differ (mode %u)\n", mode); return 1; }
            }
            std::vector<mi_g1_affine> p99(p100.begin(), p100.begin() + 99);   // sc lost one entry in the length check above
            mi_g1_jac gs = grp.MultiExpG1(p99, sc, 1), ws{};
            ref_msm_g1(p99.data(), sc.data(), 99, 0, &ws);
            if (std::memcmp(&gs, &ws, sizeof(gs)) != 0) { std::puts("MISMATCH: sharded MultiExp"); return 1; }
        }
        std::printf("OK %zu proof bytes identical\n", got.size());
        return 0;
    } catch (const groth16::Error &e) {
        std::printf("ERROR %d: %s\n", e.code, e.what());
        return 3;
    }
}
