// Fixed-base batch scalar multiplication on G1 / G2.
// Replaces gnark-crypto ecc/bn254 BatchScalarMultiplicationG1 / G2, which groth16.Setup (/root/reference/mt.go:448) uses to
// turn the evaluated polynomials A_j(tau), B_j(tau), K_j, Z_i into the points of the proving and verifying keys.
//
//   table  T[w][d] = d * 2^(8w) * base, w < 32, 1 <= d < 256, affine (512 KiB for G1: stays in L2)
//   main   thread i: s = canonical(scalars[i]); acc = sum_w T[w][byte_w(s)] with XYZZ mixed additions (<= 32), then one
//          inversion to affine.  ~700 Fp products per scalar (G1).
#include "ctx.h"
#include "curve.cuh"
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
__global__ void __launch_bounds__(64) k_fb_mul(Affine<F> *out, const Affine<F> *table, const Fr *scalars, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Fr s = fe_from_mont(scalars[i]);
    XYZZ<F> acc = XYZZ<F>::inf();
    for (u32 w = 0; w < 32; w++) {
        const u32 d = (s.l[w >> 2] >> ((w & 3) * 8)) & 255;
        if (d) xyzz_madd(acc, table[w * 256 + d], false);
    }
    out[i] = xyzz_to_affine(acc);
}

template <class F, class AffT>
static int32_t fb_run_dev(mi_ctx *ctx, const AffT *base, const mi_fr *scalars_dev, size_t n, AffT *out_dev) {
    if (!ctx || !base || ((!scalars_dev || !out_dev) && n)) return MI_EINVAL;
    if (n == 0) return MI_OK;
    MI_TRY(mi_reserve(ctx, ctx->ws[20], 32 * 256 * sizeof(Affine<F>)));
    Affine<F> b;
    std::memcpy(&b, base, sizeof(b));
    Affine<F> *table = (Affine<F> *)ctx->ws[20].p;
    hipLaunchKernelGGL(k_fb_table<F>, dim3(32 * 256 / 64), dim3(64), 0, ctx->stream, table, b);
    hipLaunchKernelGGL(k_fb_mul<F>, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, ctx->stream, (Affine<F> *)out_dev, table, (const Fr *)scalars_dev, n);
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
