// Fixed-base batch scalar multiplication on G1 / G2.
// Replaces gnark-crypto ecc/bn254 BatchScalarMultiplicationG1 / G2, which groth16.Setup (/root/reference/mt.go:448) uses to
// turn the evaluated polynomials A_j(tau), B_j(tau), K_j, Z_i into the points of the proving and verifying keys.
//
//   table  T[w][d] = d * 2^(8w) * base, w < 32, 1 <= d < 256, affine (512 KiB for G1: stays in L2)
//   main   thread i: s = canonical(scalars[i]); acc = sum_w T[w][byte_w(s)] with XYZZ mixed additions (<= 32) -> XYZZ scratch
//   affine one inversion per 16 results (batch_affine.cuh): ~8 + 24 products per point instead of ~385
// G1 (r2): the additions in nine 29-bit limbs over a table in the packed R' form (curve29.cuh, as the MSM's level-1 kernel).
// ~700 -> ~350 Fp products per scalar (G1), the products themselves 20 % cheaper.
#include "ctx.h"
#include "curve.cuh"
#include "curve29.cuh"
#include "batch_affine.cuh"
#include <cstring>

template <class F>
__global__ void __launch_bounds__(64) k_fb_table(Affine<F> *table, const Affine<F> base) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;   // t = w * 256 + d
    if (t >= 32 * 256) return;
    const u32 w = t >> 8, d = t & 255;
    if (d == 0) { table[t] = Affine<F>{F::zero(), F::zero()}; return; }
    u32 k[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    k[w >> 2] = d << ((w & 3) * 8);
    table[t] = xyzz_to_affine(xyzz_mul_256(XYZZ<F>::from_affine(base), k));
}
template <class F>
__global__ void __launch_bounds__(64) k_fb_mul(XYZZ<F> *out, const Affine<F> *table, const Fr *scalars, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Fr s = fe_from_mont(scalars[i]);
    XYZZ<F> acc = XYZZ<F>::inf();
    for (u32 w = 0; w < 32; w++) {
        const u32 d = (s.l[w >> 2] >> ((w & 3) * 8)) & 255;
        if (d) xyzz_madd(acc, table[w * 256 + d], false);
    }
    out[i] = acc;
}
// G1: the same sum in 29-bit limbs; table_rp holds both coordinates * 2^5 (the packed R' form), (0, 0) = infinity
__global__ void k_fb_table_to_rprime(G1Aff *table, u32 count) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    const G1Aff a = table[t];
    table[t] = G1Aff{fe_to_rprime_packed(a.x), fe_to_rprime_packed(a.y)};
}
__global__ void __launch_bounds__(64) k_fb_mul_g1_29(G1X *out, const G1Aff *table_rp, const Fr *scalars, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Fr s = fe_from_mont(scalars[i]);
    G1X29 acc = g1x29_inf();
    for (u32 w = 0; w < 32; w++) {
        const u32 d = (s.l[w >> 2] >> ((w & 3) * 8)) & 255;
        if (!d) continue;
        const uint4 *q4 = reinterpret_cast<const uint4 *>(table_rp + w * 256 + d);
        const uint4 q0 = q4[0], q1 = q4[1], q2 = q4[2], q3 = q4[3];
        const u32 pt[16] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
        g1x29_madd(acc, pt, false);
    }
    out[i] = g1x29_to_std(acc);
}
template <class F> struct FbG1 { static constexpr bool value = false; };
template <> struct FbG1<Fp> { static constexpr bool value = true; };

template <class F, class AffT>
static int32_t fb_run_dev(mi_ctx *ctx, const AffT *base, const mi_fr *scalars_dev, size_t n, AffT *out_dev) {
    if (!ctx || !base || ((!scalars_dev || !out_dev) && n)) return MI_EINVAL;
    if (n == 0) return MI_OK;
    MI_TRY(mi_reserve(ctx, ctx->ws[20], 32 * 256 * sizeof(Affine<F>)));
    MI_TRY(mi_reserve(ctx, ctx->ws[4], n * sizeof(XYZZ<F>)));   // the sums before their conversion
    MI_TRY(mi_reserve(ctx, ctx->ws[5], n * sizeof(F)));         // running products of the batched inversion
    Affine<F> b;
    std::memcpy(&b, base, sizeof(b));
    Affine<F> *table = (Affine<F> *)ctx->ws[20].p;
    XYZZ<F> *sums = (XYZZ<F> *)ctx->ws[4].p;
    const unsigned blocks = (unsigned)((n + 63) / 64);
    hipLaunchKernelGGL(k_fb_table<F>, dim3(32 * 256 / 64), dim3(64), 0, ctx->stream, table, b);
    if constexpr (FbG1<F>::value) {
        hipLaunchKernelGGL(k_fb_table_to_rprime, dim3(32 * 256 / 64), dim3(64), 0, ctx->stream, (G1Aff *)table, 32u * 256u);
        hipLaunchKernelGGL(k_fb_mul_g1_29, dim3(blocks), dim3(64), 0, ctx->stream, (G1X *)sums, (const G1Aff *)table, (const Fr *)scalars_dev, n);
    } else {
        hipLaunchKernelGGL(k_fb_mul<F>, dim3(blocks), dim3(64), 0, ctx->stream, sums, table, (const Fr *)scalars_dev, n);
    }
    constexpr int K = 16;
    hipLaunchKernelGGL((k_xyzz_batch_to_affine<F, K>), dim3((unsigned)(((n + K - 1) / K + 63) / 64)), dim3(64), 0, ctx->stream, (const XYZZ<F> *)sums,
                       (Affine<F> *)out_dev, (F *)ctx->ws[5].p, n);
    MI_CHECK_HIP(ctx, hipGetLastError());
    return MI_OK;
}
template <class F, class AffT>
static int32_t fb_run_host(mi_ctx *ctx, const AffT *base, const mi_fr *scalars, size_t n, AffT *out) {
    if (!ctx || !base || ((!scalars || !out) && n)) return MI_EINVAL;
    MI_TRY(mi_reserve(ctx, ctx->ws[21], n * sizeof(mi_fr) + 64));
    MI_TRY(mi_reserve(ctx, ctx->ws[22], n * sizeof(AffT) + 64));
    if (n) MI_CHECK_HIP(ctx, hipMemcpyAsync(ctx->ws[21].p, scalars, n * sizeof(mi_fr), hipMemcpyHostToDevice, ctx->stream));
    MI_TRY((fb_run_dev<F, AffT>(ctx, base, (const mi_fr *)ctx->ws[21].p, n, (AffT *)ctx->ws[22].p)));
    if (n) MI_CHECK_HIP(ctx, hipMemcpyAsync(out, ctx->ws[22].p, n * sizeof(AffT), hipMemcpyDeviceToHost, ctx->stream));
    MI_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MI_OK;
}

extern "C" {
int32_t mi_batch_scalar_mul_g1(mi_ctx *ctx, const mi_g1_affine *base, const mi_fr *scalars, size_t n, mi_g1_affine *out) {
    return fb_run_host<Fp, mi_g1_affine>(ctx, base, scalars, n, out);
}
int32_t mi_batch_scalar_mul_g1_dev(mi_ctx *ctx, const mi_g1_affine *base, const mi_fr *scalars_dev, size_t n, mi_g1_affine *out_dev) {
    return fb_run_dev<Fp, mi_g1_affine>(ctx, base, scalars_dev, n, out_dev);
}
int32_t mi_batch_scalar_mul_g2(mi_ctx *ctx, const mi_g2_affine *base, const mi_fr *scalars, size_t n, mi_g2_affine *out) {
    return fb_run_host<Fp2, mi_g2_affine>(ctx, base, scalars, n, out);
}
int32_t mi_batch_scalar_mul_g2_dev(mi_ctx *ctx, const mi_g2_affine *base, const mi_fr *scalars_dev, size_t n, mi_g2_affine *out_dev) {
    return fb_run_dev<Fp2, mi_g2_affine>(ctx, base, scalars_dev, n, out_dev);
}
}
