// computeH over the ranks of a device group (SURVEY 8e said "NTT: replicas only"; VERDICT r4 weak 3: with computeH on the lead alone a
// proof sharded over 8 GPUs is capped near 3x).  The kernels and tables of the FOUR-STEP form of a size-N = W M transform whose data
// lives in W slices of M elements, one per rank; csrc/group.hip (compute_h_sharded) moves the data and calls the local size-M transforms.
// Replaces, by behaviour, the same gnark functions as csrc/ntt.hip: fft.Domain.FFT / FFTInverse with DIF / DIT and OnCoset as computeH
// (backend/groth16/bn254/prove.go, reached from /root/reference/mt.go:496) uses them; same h, bit for bit (canonical field elements).
//
// With j = j1 M + j2 (slice j1 holds x[j1 M + j2]), k = k1 + W k2, w = the N-th root, w_W = w^M, w_M = w^W:
//   inverse, natural -> bit-reversed (DIF):   y_k1[j2] = (sum_j1 x[j1 M + j2] w_W^(-j1 k1)) * w^(-j2 k1) / W      <- k_cross_dft, mode 0
//                                             c[k1 + W k2] = FFTInverse_M(y_k1)[k2]                               <- the local size-M transform
//       coefficient k lands on rank bitrev_W(k1) at local position bitrev_M(k2): the global bit-reversed order, sliced.
//   forward on the coset, bit-reversed -> natural (DIT):   z_k1[i2] = FFT_M(c[k1 + W k2] g^(k1 + W k2))[i2]       <- scale table, local transform
//                                             e[i1 M + i2] = sum_k1 w_W^(i1 k1) (w^(i2 k1) z_k1[i2])              <- k_cross_dft, mode 1
// Between the two halves of each transform the W values that share a j2 (an i2) sit on W different ranks: one all-to-all brings them to
// the rank that owns that column range, a second one takes the results to the rank that owns the row (group.hip).  W = 2, 4, 8, 16.
#include "prove_internal.h"
#include "ntt_tile.cuh"
#include "ntt_cross.h"

struct CrossArgs {
    Fr wp[8];     // w_W^(+-e), e < W / 2
    Fr scale;     // multiplies every output of mode 0 (1 / W, den / W); one() for mode 1
    u32 cnt;      // columns this rank owns (M / W)
};

// The W values of one column in registers.  (Every loop below has a compile-time trip count and compile-time indices.)
// DIF half (the inverse transform's cross-rank step): rows in natural j1 order -> butterflies (outputs in bit-reversed k1 order: slot p
// holds k1 = bitrev(p)) -> times scale * t^k1, t = w^(-j2)
template <int LOGW>
MI_D void cross_inverse_step(Fr (&v)[1 << LOGW], const Fr &t, const Fr &scale, const Fr (&wp)[8]) {
    constexpr int W = 1 << LOGW;
    Fr pw[W];
    pw[0] = scale;
#pragma unroll
    for (int k = 1; k < W; k++) pw[k] = pw[k - 1] * t;
#pragma unroll
    for (int s = 0; s < LOGW; s++) {   // butterflies of span W >> s
        const int len = W >> s, half = len >> 1, step = 1 << s;
#pragma unroll
        for (int i = 0; i < W / 2; i++) {
            const int k = i % half, lo = (i / half) * len + k, hi = lo + half;
            const Fr u = v[lo], x = v[hi];
            v[lo] = u + x;
            v[hi] = k ? (u - x) * wp[k * step] : (u - x);
        }
    }
#pragma unroll
    for (int p = 0; p < W; p++) v[p] = v[p] * pw[bitrev_u32((u32)p, LOGW)];
}
// DIT half (the forward transform's): slots in bit-reversed k1 order -> times t^k1, t = w^(+i2) -> butterflies -> rows in natural i1 order
template <int LOGW>
MI_D void cross_forward_step(Fr (&v)[1 << LOGW], const Fr &t, const Fr (&wp)[8]) {
    constexpr int W = 1 << LOGW;
    Fr pw[W];
    pw[0] = Fr::one();
#pragma unroll
    for (int k = 1; k < W; k++) pw[k] = pw[k - 1] * t;
#pragma unroll
    for (int p = 0; p < W; p++) { const u32 k1 = bitrev_u32((u32)p, LOGW); if (k1) v[p] = v[p] * pw[k1]; }
#pragma unroll
    for (int s = LOGW - 1; s >= 0; s--) {   // spans 2, 4, ..., W
        const int len = W >> s, half = len >> 1, step = 1 << s;
#pragma unroll
        for (int i = 0; i < W / 2; i++) {
            const int k = i % half, lo = (i / half) * len + k, hi = lo + half;
            const Fr u = v[lo], x = k ? v[hi] * wp[k * step] : v[hi];
            v[lo] = u + x;
            v[hi] = u - x;
        }
    }
}
// Y: W rows of cnt elements (row = the rank the values came from).  One thread per column; mode 0 = inverse step, 1 = forward step.
template <int LOGW>
__global__ void __launch_bounds__(256) k_cross_dft(Fr *Y, const Fr *tw, CrossArgs a, int mode) {
    constexpr int W = 1 << LOGW;
    const u32 j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= a.cnt) return;
    Fr v[W];
#pragma unroll
    for (int p = 0; p < W; p++) v[p] = Y[(size_t)p * a.cnt + j];
    const Fr t = tw[j];
    if (mode == 0) cross_inverse_step<LOGW>(v, t, a.scale, a.wp);
    else cross_forward_step<LOGW>(v, t, a.wp);
#pragma unroll
    for (int p = 0; p < W; p++) Y[(size_t)p * a.cnt + j] = v[p];
}
// computeH's middle on the columns: the forward step of a and of b, the product a b, the inverse step of the last transform -- one
// read of the two column blocks and one write instead of three kernels over them (Ya <- result).
template <int LOGW>
__global__ void __launch_bounds__(256) k_cross_mid(Fr *Ya, const Fr *Yb, const Fr *tw_fwd, const Fr *tw_inv, CrossArgs f, CrossArgs inv) {
    constexpr int W = 1 << LOGW;
    const u32 j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= f.cnt) return;
    Fr va[W], vb[W];
#pragma unroll
    for (int p = 0; p < W; p++) { va[p] = Ya[(size_t)p * f.cnt + j]; vb[p] = Yb[(size_t)p * f.cnt + j]; }
    const Fr t = tw_fwd[j];
    cross_forward_step<LOGW>(va, t, f.wp);
    cross_forward_step<LOGW>(vb, t, f.wp);
#pragma unroll
    for (int p = 0; p < W; p++) va[p] = va[p] * vb[p];
    cross_inverse_step<LOGW>(va, tw_inv[j], inv.scale, inv.wp);
#pragma unroll
    for (int p = 0; p < W; p++) Ya[(size_t)p * f.cnt + j] = va[p];
}

// out[j] = c * base^(first + j)
__global__ void k_cross_pow(Fr *out, u32 count, Fr base, Fr c, u64 first) {
    const u32 j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count) return;
    u64 e = first + j;
    Fr acc = c, b = base;
    while (e) { if (e & 1) acc = acc * b; b = fe_sqr(b); e >>= 1; }
    out[j] = acc;
}
// out[t] = c * base^bitrev_M(t): the coset factor of the coefficient that sits at local position t
__global__ void k_cross_scale_table(Fr *out, u32 log_m, Fr base, Fr c) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (1u << log_m)) return;
    u32 e = bitrev_u32(t, log_m);
    Fr acc = c, b = base;
    while (e) { if (e & 1) acc = acc * b; b = fe_sqr(b); e >>= 1; }
    out[t] = acc;
}
// z = x * y, z = x * y - w (one pass each; the last step of the sharded computeH is the second)
__global__ void k_cross_mul(Fr *z, const Fr *x, const Fr *y, const Fr *w, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr r = x[i] * y[i];
    if (w) r = r - w[i];
    z[i] = r;
}

static Fr host_pow(Fr b, u64 e) {
    Fr acc = Fr::one();
    while (e) { if (e & 1) acc = acc * b; b = fe_sqr(b); e >>= 1; }
    return acc;
}
static Fr cross_domain_generator(u32 log_n) {   // fft.NewDomain: Generator = root^(2^(28 - log_n)), as csrc/ntt.hip
    Fr t;
    const u64 lim[4] = {0x9bd61b6e725b19f0ull, 0x402d111e41112ed4ull, 0x00e0a7eb8ef62abcull, 0x2a3c09f0a58a7e85ull};
    for (int i = 0; i < 4; i++) { t.l[2 * i] = (u32)lim[i]; t.l[2 * i + 1] = (u32)(lim[i] >> 32); }
    Fr g = fe_to_mont(t);
    for (u32 k = log_n; k < 28; k++) g = fe_sqr(g);
    return g;
}

void mi_cross_tables_free(CrossNttTables *t) {
    for (void **p : {&t->tw_inv, &t->tw_fwd, &t->s_fwd, &t->s_inv}) if (*p) { (void)hipFree(*p); *p = nullptr; }
    t->log_n = 0;
}
// rank's tables for (log_n, log_w): twiddles w^(-+ j2) of its columns, coset factors g^k and den g^-k of the coefficients it holds
int32_t mi_cross_tables_build(mi_ctx *ctx, u32 log_n, u32 log_w, u32 rank, CrossNttTables *t) {
    if (t->log_n == log_n && t->log_w == log_w && t->rank == rank) return MI_OK;
    mi_cross_tables_free(t);
    if (log_w < 1 || log_w > 4 || log_n < 2 * log_w || log_n > 28) MI_FAIL(ctx, MI_EINVAL, "sharded computeH: needs 2 <= ranks <= 16 (a power of two) and N >= ranks^2");
    const u32 W = 1u << log_w, log_m = log_n - log_w, cnt = 1u << (log_m - log_w);
    const size_t M = (size_t)1 << log_m;
    MI_CHECK_HIP(ctx, hipMalloc(&t->tw_inv, sizeof(Fr) * cnt));
    MI_CHECK_HIP(ctx, hipMalloc(&t->tw_fwd, sizeof(Fr) * cnt));
    MI_CHECK_HIP(ctx, hipMalloc(&t->s_fwd, sizeof(Fr) * M));
    MI_CHECK_HIP(ctx, hipMalloc(&t->s_inv, sizeof(Fr) * M));
    const Fr w = cross_domain_generator(log_n), wi = fe_inv(w), g = fe_from_u32<FrParams>(5), gi = fe_inv(g);
    Fr gn = g;
    for (u32 k = 0; k < log_n; k++) gn = fe_sqr(gn);
    const Fr den = fe_inv(gn - Fr::one());
    const u64 first = (u64)rank * cnt;   // this rank's first column
    hipLaunchKernelGGL(k_cross_pow, dim3((cnt + 127) / 128), dim3(128), 0, ctx->stream, (Fr *)t->tw_inv, cnt, wi, Fr::one(), first);
    hipLaunchKernelGGL(k_cross_pow, dim3((cnt + 127) / 128), dim3(128), 0, ctx->stream, (Fr *)t->tw_fwd, cnt, w, Fr::one(), first);
    // coefficients on this rank: k = k1 + W k2 with k1 = bitrev_W(rank), local position bitrev_M(k2)
    const u32 k1 = bitrev_u32(rank, log_w);
    const unsigned blocks = (unsigned)((M + 255) / 256);
    hipLaunchKernelGGL(k_cross_scale_table, dim3(blocks), dim3(256), 0, ctx->stream, (Fr *)t->s_fwd, log_m, host_pow(g, W), host_pow(g, k1));
    hipLaunchKernelGGL(k_cross_scale_table, dim3(blocks), dim3(256), 0, ctx->stream, (Fr *)t->s_inv, log_m, host_pow(gi, W), den * host_pow(gi, k1));
    MI_CHECK_HIP(ctx, hipGetLastError());
    Fr nw = Fr::zero();
    nw.l[0] = W;
    t->w_inv_scale = fe_inv(fe_to_mont(nw));   // 1 / W
    t->den = den;
    t->log_n = log_n; t->log_w = log_w; t->rank = rank;
    return MI_OK;
}
static void cross_args(const CrossNttTables &t, int mode, bool den_scale, CrossArgs *a) {
    const u32 W = 1u << t.log_w;
    const Fr ww = cross_domain_generator(t.log_w), base = mode == 0 ? fe_inv(ww) : ww;
    Fr p = Fr::one();
    for (u32 e = 0; e < 8; e++) { a->wp[e] = p; if (e + 1 < W / 2) p = p * base; }
    a->scale = mode == 0 ? (den_scale ? t.w_inv_scale * t.den : t.w_inv_scale) : Fr::one();
    a->cnt = 1u << (t.log_n - 2 * t.log_w);
}
// in place on Y (W rows of M / W columns) on stream st; den_scale: mode 0's outputs carry den / W instead of 1 / W
int32_t mi_cross_dft(mi_ctx *ctx, hipStream_t st, void *Y, const CrossNttTables &t, int mode, bool den_scale) {
    CrossArgs a;
    cross_args(t, mode, den_scale, &a);
    const u32 cnt = a.cnt;
    const Fr *tw = (const Fr *)(mode == 0 ? t.tw_inv : t.tw_fwd);
    const dim3 grid((cnt + 255) / 256), block(256);
    switch (t.log_w) {
    case 1: hipLaunchKernelGGL(k_cross_dft<1>, grid, block, 0, st, (Fr *)Y, tw, a, mode); break;
    case 2: hipLaunchKernelGGL(k_cross_dft<2>, grid, block, 0, st, (Fr *)Y, tw, a, mode); break;
    case 3: hipLaunchKernelGGL(k_cross_dft<3>, grid, block, 0, st, (Fr *)Y, tw, a, mode); break;
    default: hipLaunchKernelGGL(k_cross_dft<4>, grid, block, 0, st, (Fr *)Y, tw, a, mode); break;
    }
    MI_CHECK_HIP(ctx, hipGetLastError());
    return MI_OK;
}
// Ya <- inverse step(forward step(Ya) o forward step(Yb)), in place on stream st (k_cross_mid)
int32_t mi_cross_mid(mi_ctx *ctx, hipStream_t st, void *Ya, const void *Yb, const CrossNttTables &t) {
    CrossArgs f, inv;
    cross_args(t, 1, false, &f); cross_args(t, 0, false, &inv);
    const dim3 grid((f.cnt + 255) / 256), block(256);
    const Fr *tf = (const Fr *)t.tw_fwd, *ti = (const Fr *)t.tw_inv;
    switch (t.log_w) {
    case 1: hipLaunchKernelGGL(k_cross_mid<1>, grid, block, 0, st, (Fr *)Ya, (const Fr *)Yb, tf, ti, f, inv); break;
    case 2: hipLaunchKernelGGL(k_cross_mid<2>, grid, block, 0, st, (Fr *)Ya, (const Fr *)Yb, tf, ti, f, inv); break;
    case 3: hipLaunchKernelGGL(k_cross_mid<3>, grid, block, 0, st, (Fr *)Ya, (const Fr *)Yb, tf, ti, f, inv); break;
    default: hipLaunchKernelGGL(k_cross_mid<4>, grid, block, 0, st, (Fr *)Ya, (const Fr *)Yb, tf, ti, f, inv); break;
    }
    MI_CHECK_HIP(ctx, hipGetLastError());
    return MI_OK;
}
// z = x o y (- w when w != null), n elements, on stream st
int32_t mi_cross_mul(mi_ctx *ctx, hipStream_t st, void *z, const void *x, const void *y, const void *w, size_t n) {
    if (n) hipLaunchKernelGGL(k_cross_mul, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (Fr *)z, (const Fr *)x, (const Fr *)y, (const Fr *)w, n);
    MI_CHECK_HIP(ctx, hipGetLastError());
    return MI_OK;
}
