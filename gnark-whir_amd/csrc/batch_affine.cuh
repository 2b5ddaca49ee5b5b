// XYZZ -> affine for MANY points with one field inversion per thread (Montgomery's trick), and the doubling chain of the fixed-base
// window tables built on it.  An inversion is ~380 Fp products (Fermat); a point conversion by itself (xyzz_to_affine) is that plus
// five.  Here thread t walks K consecutive points twice: forwards it multiplies their ZZZ up, parking the running product of the
// points before each one in a scratch array (one field element per point); it inverts the total ONCE; backwards it peels one
// 1 / ZZZ_j per point off the inverse (two products) and finishes the point (1 / ZZ = ZZ^2 / ZZZ^2, x = X / ZZ, y = Y / ZZZ: five
// products): ~8 + 380 / K products per point, the same field elements (an inverse is unique), so the same bytes.
// Used by the fixed-base batch scalar multiplication (fixed_base.hip) and by mi_msm_precompute (the window tables of
// mi_pk_load: one conversion per point per window).
#pragma once
#include "curve.cuh"

template <class F, int K>
__global__ void __launch_bounds__(64) k_xyzz_batch_to_affine(const XYZZ<F> *in, Affine<F> *out, F *prefix, size_t n) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x, b = t * K;
    if (b >= n) return;
    const u32 m = n - b < (size_t)K ? (u32)(n - b) : (u32)K;
    F run = F::one();
    for (u32 j = 0; j < m; j++) {
        const F zzz = in[b + j].zzz;
        prefix[b + j] = run;
        if (!zzz.is_zero()) run = run * zzz;   // the point at infinity (ZZ = ZZZ = 0) takes no part in the product
    }
    F inv = fe_inv(run);
    for (u32 j = m; j-- > 0;) {
        const XYZZ<F> p = in[b + j];
        if (p.is_inf()) { out[b + j] = Affine<F>{F::zero(), F::zero()}; continue; }
        const F izzz = inv * prefix[b + j];
        inv = inv * p.zzz;
        const F izz = fe_sqr(izzz) * fe_sqr(p.zz);
        out[b + j] = Affine<F>{p.x * izz, p.y * izzz};
    }
}
template <class F>
__global__ void __launch_bounds__(64) k_xyzz_from_affine(XYZZ<F> *state, Affine<F> *first_window, const Affine<F> *base, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Affine<F> p = base[i];
    first_window[i] = p;
    state[i] = XYZZ<F>::from_affine(p);
}
// state[i] <- 2^c * state[i]
template <class F>
__global__ void __launch_bounds__(64) k_xyzz_dbl_c(XYZZ<F> *state, size_t n, u32 c) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    XYZZ<F> acc = state[i];
    for (u32 k = 0; k < c; k++) acc = xyzz_dbl(acc);
    state[i] = acc;
}
// the window copies pre[w][i] = 2^(c w) * base[i] (msm2_core.cuh msm2_precompute_body: same points, same bytes) with batched conversions:
// state (n XYZZ) and prefix (n field elements) are scratch
template <class F>
static void launch_precompute_batched(hipStream_t st, const void *base, void *pre, uint32_t n, uint32_t c, uint32_t nwin, void *state, void *prefix) {
    constexpr int K = 16;
    const unsigned blocks = (n + 63) / 64, cblocks = (unsigned)(((size_t)n + K - 1) / K + 63) / 64;
    hipLaunchKernelGGL(k_xyzz_from_affine<F>, dim3(blocks), dim3(64), 0, st, (XYZZ<F> *)state, (Affine<F> *)pre, (const Affine<F> *)base, (size_t)n);
    for (uint32_t w = 1; w < nwin; w++) {
        hipLaunchKernelGGL(k_xyzz_dbl_c<F>, dim3(blocks), dim3(64), 0, st, (XYZZ<F> *)state, (size_t)n, c);
        hipLaunchKernelGGL((k_xyzz_batch_to_affine<F, K>), dim3(cblocks), dim3(64), 0, st, (const XYZZ<F> *)state, (Affine<F> *)pre + (size_t)w * n, (F *)prefix, (size_t)n);
    }
}
