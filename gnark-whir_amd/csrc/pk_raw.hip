// Proving-key ingestion from gnark's on-disk format: the stream gnark v0.11.0's groth16 bn254
// `(*ProvingKey).WriteRawTo` writes.  The reference regenerates its key on every run (groth16.Setup, /root/reference/mt.go:448);
// a deployment loads one, and a CPU / GPU parity run needs both provers on the SAME key.
//
// LAYOUT RECALLED, UNVERIFIED (no Go toolchain, gnark absent from the image): restated from the published behaviour of gnark
// backend/groth16/bn254/marshal.go and gnark-crypto ecc/bn254/marshal.go (Encoder with RawEncoding), fr/fft Domain.WriteTo,
// fr/pedersen ProvingKey.WriteRawTo; spelled out field by field in oracle/pk_raw.py, which writes the same layout for the tests.
// The parser is strict -- every count is cross-checked (points vs infinity masks, stream length, flag bits, the domain's
// constants against log_n) -- so a SHIFTED or re-ordered layout shows up as MI_EINVAL on first contact.  ONE thing no check here can
// see: the bit order inside the packed []bool masks (taken as MSB-first, `7 - j % 8`).  A population count is the same under
// either order, so the other order would pass every check and pair the A / B points with the wrong wires: wrong proofs, silently.
// The entry point is therefore EXPERIMENTAL until one real gnark v0.11.0 WriteRawTo fixture (go/mi355x/cmd/dumpfixture) has been
// loaded and a proof from it verified; include/mi355x_groth16.h and INTEGRATION.md say the same.
//
// Raw points are big-endian CANONICAL coordinates; the device wants little-endian Montgomery limbs.  The conversion is the one
// data-parallel piece (one point per thread: byte swap, flag bits, fe_to_mont) and runs on the GPU over the uploaded bytes;
// the variable-length header walk is host code (csrc/pk_raw_inspect.hip: no HIP in it, so that the CPU build with sanitizers covers it).
#include "prove_internal.h"
#include <cstring>
#include <vector>


// one field element: 32 bytes big-endian canonical -> 8 x u32 little-endian Montgomery; top_mask clears the encoder's flag bits
template <class P>
MI_D Fe<P> fe_from_be(const uint8_t *b, uint8_t top_mask) {
    Fe<P> t;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint8_t *q = b + 28 - 4 * i;
        t.l[i] = ((u32)q[0] << 24) | ((u32)q[1] << 16) | ((u32)q[2] << 8) | (u32)q[3];
    }
    t.l[7] &= ((u32)top_mask << 24) | 0x00ffffffu;
    return fe_to_mont(t);
}
__global__ void k_raw_to_g1(G1Aff *dst, const uint8_t *src, size_t n, u32 *bad) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t *b = src + 64 * i;
    const u32 flags = b[0] >> 6;   // 00 uncompressed, 01 uncompressed infinity; 10 / 11 are compressed forms: not a raw stream
    if (flags >= 2) { atomicOr(bad, 1u); return; }
    G1Aff a;
    if (flags == 1) { a.x = Fp::zero(); a.y = Fp::zero(); }
    else { a.x = fe_from_be<FpParams>(b, 0x3f); a.y = fe_from_be<FpParams>(b + 32, 0xff); }
    dst[i] = a;
}
__global__ void k_raw_to_g2(G2Aff *dst, const uint8_t *src, size_t n, u32 *bad) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t *b = src + 128 * i;   // X.A1 | X.A0 | Y.A1 | Y.A0
    const u32 flags = b[0] >> 6;
    if (flags >= 2) { atomicOr(bad, 1u); return; }
    G2Aff a;
    if (flags == 1) { a.x = Fp2::zero(); a.y = Fp2::zero(); }
    else {
        a.x.a1 = fe_from_be<FpParams>(b, 0x3f); a.x.a0 = fe_from_be<FpParams>(b + 32, 0xff);
        a.y.a1 = fe_from_be<FpParams>(b + 64, 0xff); a.y.a0 = fe_from_be<FpParams>(b + 96, 0xff);
    }
    dst[i] = a;
}

extern "C" {

// nb_public and the wires removed from K (committed + commitment wires) come from the constraint system, not from the key file
// (r1cs.GetNbPublicVariables(), CommitmentInfo): the caller passes them as for mi_pk_load.  ped_out (may be null) receives the
// keys' Pedersen commitment keys, device-resident, for mi_pedersen_*.
int32_t mi_pk_load_raw(mi_ctx *ctx, const uint8_t *buf, size_t len, uint32_t nb_public, const uint32_t *committed_wires, size_t n_committed,
                       mi_pk **out, mi_pedersen_pk **ped_out, uint32_t *n_ped_out) {
    if (!ctx || !buf || !out) return MI_EINVAL;
    *out = nullptr;
    if (n_ped_out) *n_ped_out = 0;
    mi_pk_raw_info in;
    if (mi_pk_raw_inspect(buf, len, &in) != MI_OK) MI_FAIL(ctx, MI_EINVAL, "pk raw: the stream does not have the layout of gnark v0.11.0 ProvingKey.WriteRawTo (see csrc/pk_raw.hip)");
    // the domain's constants must be the ones fft.NewDomain derives for this size (a cheap canary for a shifted layout)
    {
        Fr want_gen = Fr::zero();
        const uint8_t *g = buf + 8 + 32;   // Generator
        for (int i = 0; i < 8; i++) { const uint8_t *q = g + 28 - 4 * i; want_gen.l[i] = ((u32)q[0] << 24) | ((u32)q[1] << 16) | ((u32)q[2] << 8) | q[3]; }
        Fr gen = fe_to_mont(want_gen), acc = gen;
        for (u32 k = 0; k < in.log_n; k++) acc = fe_sqr(acc);
        Fr half = gen;
        for (u32 k = 0; k + 1 < in.log_n; k++) half = fe_sqr(half);
        if (acc != Fr::one() || (in.log_n && half == Fr::one())) MI_FAIL(ctx, MI_EINVAL, "pk raw: Domain.Generator is not a primitive 2^log_n-th root of unity");
    }
    // infinity masks: bit-packed -> one byte per wire
    std::vector<uint8_t> ia(in.nb_wires), ib(in.nb_wires);
    for (uint64_t j = 0; j < in.nb_wires; j++) {
        ia[j] = (buf[in.off_infinity_a + j / 8] >> (7 - j % 8)) & 1;
        ib[j] = (buf[in.off_infinity_b + j / 8] >> (7 - j % 8)) & 1;
    }
    // upload the stream once; convert section by section on the device
    void *raw = nullptr;
    u32 *bad = nullptr;
    MI_CHECK_HIP(ctx, hipMalloc(&raw, len + 64));
    void *arrays[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    G1Aff small1[3];
    G2Aff small2[2];
    void *d_small = nullptr;
    int32_t rc = MI_OK;
    auto body = [&]() -> int32_t {
        MI_CHECK_HIP(ctx, hipMalloc((void **)&bad, 4));
        MI_CHECK_HIP(ctx, hipMemsetAsync(bad, 0, 4, ctx->stream));
        MI_CHECK_HIP(ctx, hipMemcpyAsync(raw, buf, len, hipMemcpyHostToDevice, ctx->stream));
        MI_CHECK_HIP(ctx, hipMalloc(&d_small, 3 * 64 + 2 * 128));
        auto conv1 = [&](void **dst, uint64_t off, uint64_t n) -> int32_t {
            if (!*dst) MI_CHECK_HIP(ctx, hipMalloc(dst, n ? n * 64 : 64));
            if (n) hipLaunchKernelGGL(k_raw_to_g1, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, ctx->stream, (G1Aff *)*dst, (const uint8_t *)raw + off, (size_t)n, bad);
            MI_CHECK_HIP(ctx, hipGetLastError());
            return MI_OK;
        };
        MI_TRY(conv1(&arrays[0], in.off_g1_a, in.n_g1_a));
        MI_TRY(conv1(&arrays[1], in.off_g1_b, in.n_g1_b));
        MI_TRY(conv1(&arrays[2], in.off_g1_k, in.n_g1_k));
        MI_TRY(conv1(&arrays[3], in.off_g1_z, in.n_g1_z));
        MI_CHECK_HIP(ctx, hipMalloc(&arrays[4], in.n_g2_b ? in.n_g2_b * 128 : 128));
        if (in.n_g2_b) hipLaunchKernelGGL(k_raw_to_g2, dim3((unsigned)((in.n_g2_b + 127) / 128)), dim3(128), 0, ctx->stream, (G2Aff *)arrays[4], (const uint8_t *)raw + in.off_g2_b, (size_t)in.n_g2_b, bad);
        void *ds1 = d_small;
        MI_TRY(conv1(&ds1, in.off_alpha1, 3));
        hipLaunchKernelGGL(k_raw_to_g2, dim3(1), dim3(128), 0, ctx->stream, (G2Aff *)((char *)d_small + 192), (const uint8_t *)raw + in.off_beta2, (size_t)2, bad);
        MI_CHECK_HIP(ctx, hipGetLastError());
        u32 bad_h = 0;
        MI_CHECK_HIP(ctx, hipMemcpyAsync(small1, d_small, 192, hipMemcpyDeviceToHost, ctx->stream));
        MI_CHECK_HIP(ctx, hipMemcpyAsync(small2, (char *)d_small + 192, 256, hipMemcpyDeviceToHost, ctx->stream));
        MI_CHECK_HIP(ctx, hipMemcpyAsync(&bad_h, bad, 4, hipMemcpyDeviceToHost, ctx->stream));
        MI_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (bad_h) MI_FAIL(ctx, MI_EINVAL, "pk raw: a point carries compressed-encoding flag bits (this is a WriteTo stream, not WriteRawTo?)");
        mi_pk_desc d;
        std::memset(&d, 0, sizeof(d));
        d.log_n = in.log_n; d.nb_public = nb_public; d.nb_wires = in.nb_wires;
        d.g1_a = (const mi_g1_affine *)arrays[0]; d.n_g1_a = in.n_g1_a;
        d.g1_b = (const mi_g1_affine *)arrays[1]; d.n_g1_b = in.n_g1_b;
        d.g1_k = (const mi_g1_affine *)arrays[2]; d.n_g1_k = in.n_g1_k;
        d.g1_z = (const mi_g1_affine *)arrays[3]; d.n_g1_z = in.n_g1_z;
        d.g2_b = (const mi_g2_affine *)arrays[4]; d.n_g2_b = in.n_g2_b;
        std::memcpy(&d.alpha1, &small1[0], 64); std::memcpy(&d.beta1, &small1[1], 64); std::memcpy(&d.delta1, &small1[2], 64);
        std::memcpy(&d.beta2, &small2[0], 128); std::memcpy(&d.delta2, &small2[1], 128);
        d.infinity_a = ia.data(); d.infinity_b = ib.data();
        d.committed_wires = committed_wires; d.n_committed = n_committed;
        bool took = false;
        const int32_t lr = mi_pk_load_range(ctx, &d, out, true, nullptr, /*adopt=*/true, &took);
        if (took) for (void *&a : arrays) a = nullptr;   // the key owns them now (it has released them itself if it failed after taking them)
        MI_TRY(lr);
        // Pedersen commitment keys
        if (ped_out) {
            for (uint32_t k = 0; k < in.n_commitment_keys; k++) {
                void *basis = nullptr, *bes = nullptr;
                int32_t r = conv1(&basis, in.off_basis[k], in.n_basis[k]);
                if (r == MI_OK) r = conv1(&bes, in.off_basis_exp_sigma[k], in.n_basis[k]);
                if (r == MI_OK) r = mi_pedersen_pk_adopt(ctx, basis, bes, (size_t)in.n_basis[k], &ped_out[k]);
                if (r != MI_OK) { if (basis) (void)hipFree(basis); if (bes) (void)hipFree(bes); return r; }
                if (n_ped_out) *n_ped_out = k + 1;
            }
            MI_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        }
        return MI_OK;
    };
    rc = body();
    (void)hipStreamSynchronize(ctx->stream);
    for (void *a : arrays) if (a) (void)hipFree(a);
    if (raw) (void)hipFree(raw);
    if (bad) (void)hipFree(bad);
    if (d_small) (void)hipFree(d_small);
    if (rc != MI_OK && *out) { mi_pk_free(ctx, *out); *out = nullptr; }
    return rc;
}

}  // extern "C"
