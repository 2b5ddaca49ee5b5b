// Device utilities of the C-ABI: elementwise field / curve ops for parity tests, the seeded
// synthetic-workload generators of SURVEY.md 8d (same definition as oracle/groth16_ref.c so the
// two can be compared bit for bit), and VALU / modular-multiply throughput probes.
#include "ctx.h"
#include "curve.cuh"

// ---------------------------------------------------------------- seeded PRNG (counter based)
MI_HD u64 sm64(u64 z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
MI_HD u64 rnd(u64 seed, u64 idx, u64 k) { return sm64(seed ^ sm64(idx * 8 + k)); }
template <class P>
MI_HD Fe<P> rnd_fe(u64 seed, u64 idx) {  // 254-bit value, one conditional subtraction
    Fe<P> z;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        u64 v = rnd(seed, idx, k);
        if (k == 3) v &= 0x3FFFFFFFFFFFFFFFull;
        z.l[2 * k] = (u32)v;
        z.l[2 * k + 1] = (u32)(v >> 32);
    }
    return fe_reduce_once(z);
}
MI_HD Fp fp_from_limbs64(u64 a, u64 b, u64 c, u64 d) {
    Fp t;
    t.l[0] = (u32)a; t.l[1] = (u32)(a >> 32); t.l[2] = (u32)b; t.l[3] = (u32)(b >> 32);
    t.l[4] = (u32)c; t.l[5] = (u32)(c >> 32); t.l[6] = (u32)d; t.l[7] = (u32)(d >> 32);
    return fe_to_mont(t);
}

__global__ void k_gen_scalars(Fr *out, size_t n, u64 seed, int dist) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr z;
    if (dist == MI_DIST_UNIFORM) z = rnd_fe<FrParams>(seed, i);
    else {
        // MI_DIST_WHIR: per cent 45 / 25 / 5; MI_DIST_MIX(bit, byte, u64): the caller's per-mille thresholds; the rest uniform
        const bool mix = (dist & MI_DIST_MIX_FLAG) != 0;
        const u64 t0 = mix ? (u64)((dist >> 20) & 1023) : 45, t1 = t0 + (mix ? (u64)((dist >> 10) & 1023) : 25), t2 = t1 + (mix ? (u64)(dist & 1023) : 5);
        u64 u = rnd(seed, i, 4) % (mix ? 1000 : 100);
        z = Fr::zero();
        u64 v = rnd(seed, i, 5);
        if (u < t0) z.l[0] = (u32)(v & 1);
        else if (u < t1) z.l[0] = (u32)(v & 255);
        else if (u < t2) { z.l[0] = (u32)v; z.l[1] = (u32)(v >> 32); }
        else z = rnd_fe<FrParams>(seed, i);
    }
    out[i] = fe_to_mont(z);
}
__global__ void k_gen_g1(G1Aff *out, size_t n, u64 seed) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u32 e[8] = {0xb61f3f52u, 0x4f082305u, 0x5a1c72a3u, 0x65e05aa4u, 0xa0605617u, 0x6e14116du, 0xb84c680au, 0x0c19139cu};  // (q+1)/4
    Fp xc = rnd_fe<FpParams>(seed, i);
    Fp b = fe_from_u32<FpParams>(3);
    Fp x, y;
    for (;;) {
        x = fe_to_mont(xc);
        Fp rhs = fe_sqr(x) * x + b;
        y = fe_pow(rhs, e);
        if (fe_sqr(y) == rhs) break;
        Fp one = Fp::zero();
        one.l[0] = 1;
        Fp t;
        fe_add_raw(t, xc, one);
        xc = fe_reduce_once(t);
    }
    if (rnd(seed, i, 5) & 1) y = fe_neg(y);
    out[i] = G1Aff{x, y};
}
__global__ void k_gen_g2(G2Aff *out, size_t n, u64 seed) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    G2Aff g;
    g.x.a0 = fp_from_limbs64(0x46debd5cd992f6edull, 0x674322d4f75edaddull, 0x426a00665e5c4479ull, 0x1800deef121f1e76ull);
    g.x.a1 = fp_from_limbs64(0x97e485b7aef312c2ull, 0xf1aa493335a9e712ull, 0x7260bfb731fb5d25ull, 0x198e9393920d483aull);
    g.y.a0 = fp_from_limbs64(0x4ce6cc0166fa7daaull, 0xe3d1e7690c43d37bull, 0x4aab71808dcb408full, 0x12c85ea5db8c6debull);
    g.y.a1 = fp_from_limbs64(0x55acdadcd122975bull, 0xbc4b313370b38ef3ull, 0xec9e99ad690c3395ull, 0x090689d0585ff075ull);
    u64 k = rnd(seed, i, 0) | 1;
    G2X acc = G2X::inf();
    for (int bit = 63; bit >= 0; bit--) {
        acc = xyzz_dbl(acc);
        if ((k >> bit) & 1) xyzz_madd(acc, g, false);
    }
    out[i] = xyzz_to_affine(acc);
}

template <class P>
__global__ void k_field_op(int op, Fe<P> *z, const Fe<P> *x, const Fe<P> *y, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fe<P> a = x[i], b = y ? y[i] : x[i], r;
    switch (op) {
    case 0: r = a + b; break;
    case 1: r = a - b; break;
    case 2: r = a * b; break;
    case 3: r = fe_inv(a); break;
    case 4: r = fe_to_mont(a); break;
    case 5: r = fe_from_mont(a); break;
    case 6: r = fe_mul2_add(a, b, b, a); break;   // 2ab/R   (dual-product columns)
    case 7: r = fe_mul_sub(a, b, b, b); break;    // (ab - b^2)/R
    default: r = fe_sqr(a); break;
    }
    z[i] = r;
}
template <class F>
__global__ void k_ec_add(Affine<F> *out, const Affine<F> *a, const Affine<F> *b, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    XYZZ<F> acc = XYZZ<F>::from_affine(a[i]);
    xyzz_madd(acc, b[i], false);
    // second formula family too: (a + b) + b - b through the full XYZZ add / negated madd
    XYZZ<F> q = XYZZ<F>::from_affine(b[i]);
    xyzz_add(acc, q);
    xyzz_madd(acc, b[i], true);
    out[i] = xyzz_to_affine(acc);
}

// dependent chain of modular products per thread: modmul throughput probe
template <class P>
__global__ void k_bench_modmul(Fe<P> *scratch, u32 iters) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    Fe<P> x = rnd_fe<P>(1, i), y = rnd_fe<P>(2, i);
    for (u32 k = 0; k < iters; k++) {
        x = x * y;
        y = y * x;
    }
    if (x.l[0] == 0x12345678u && y.l[3] == 77u) scratch[i & 1023] = x;  // keep live, ~never taken
}
// raw VALU probes: kind 0 = 32x32+64 mad (v_mad_u64_u32), 1 = v_mul_lo+hi u32, 2 = fma f64,
// 3 = 24-bit mad, 4 = add/addc chain
__global__ void __launch_bounds__(64) k_bench_gather(const uint4 *tab, u64 mask, u32 iters, uint4 *out) {
    const u32 tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint4 acc = make_uint4(tid, 0, 0, 0);
    u64 h = tid * 0x9E3779B97F4A7C15ull;
    for (u32 it = 0; it < iters; it++) {
        h = h * 6364136223846793005ull + 1442695040888963407ull + acc.x;   // the next address depends on this gather, as in a bucket's chain
        const uint4 *q = tab + ((h >> 20) & mask) * 4;
        const uint4 a = q[0], b = q[1], c = q[2], d = q[3];
        acc.x ^= a.x ^ b.y ^ c.z ^ d.w; acc.y += a.y + d.x; acc.z ^= b.z; acc.w += c.w;
    }
    out[tid & 63] = acc;
}
__global__ void k_bench_valu(int kind, u32 iters, u64 *sink) {
    u32 t = threadIdx.x + blockIdx.x * blockDim.x;
    u64 a0 = t, a1 = t + 1, a2 = t + 2, a3 = t + 3, a4 = t + 4, a5 = t + 5, a6 = t + 6, a7 = t + 7;
    u32 m = t * 2654435761u + 12345u, k2 = t ^ 0x9e3779b9u;
    if (kind == 0) {
        for (u32 i = 0; i < iters; i++) {
            a0 = (u64)(u32)a1 * m + a0; a1 = (u64)(u32)a2 * m + a1; a2 = (u64)(u32)a3 * m + a2; a3 = (u64)(u32)a4 * m + a3;
            a4 = (u64)(u32)a5 * m + a4; a5 = (u64)(u32)a6 * m + a5; a6 = (u64)(u32)a7 * m + a6; a7 = (u64)(u32)a0 * m + a7;
        }
    } else if (kind == 1) {
        u32 b0 = (u32)a0, b1 = (u32)a1, b2 = (u32)a2, b3 = (u32)a3, b4 = (u32)a4, b5 = (u32)a5, b6 = (u32)a6, b7 = (u32)a7;
        for (u32 i = 0; i < iters; i++) {
            b0 = b1 * m + __umulhi(b0, k2); b1 = b2 * m + __umulhi(b1, k2); b2 = b3 * m + __umulhi(b2, k2); b3 = b4 * m + __umulhi(b3, k2);
            b4 = b5 * m + __umulhi(b4, k2); b5 = b6 * m + __umulhi(b5, k2); b6 = b7 * m + __umulhi(b6, k2); b7 = b0 * m + __umulhi(b7, k2);
        }
        a0 = b0 ^ b1 ^ b2 ^ b3 ^ b4 ^ b5 ^ b6 ^ b7;
    } else if (kind == 2) {
        double d0 = t, d1 = t + 1, d2 = t + 2, d3 = t + 3, d4 = t + 4, d5 = t + 5, d6 = t + 6, d7 = t + 7, mm = 1.0000001, cc = 0.5;
        for (u32 i = 0; i < iters; i++) {
            d0 = __builtin_fma(d0, mm, cc); d1 = __builtin_fma(d1, mm, cc); d2 = __builtin_fma(d2, mm, cc); d3 = __builtin_fma(d3, mm, cc);
            d4 = __builtin_fma(d4, mm, cc); d5 = __builtin_fma(d5, mm, cc); d6 = __builtin_fma(d6, mm, cc); d7 = __builtin_fma(d7, mm, cc);
        }
        a0 = (u64)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);
    } else if (kind == 3) {
        u32 b0 = (u32)a0, b1 = (u32)a1, b2 = (u32)a2, b3 = (u32)a3, b4 = (u32)a4, b5 = (u32)a5, b6 = (u32)a6, b7 = (u32)a7;
        for (u32 i = 0; i < iters; i++) {
            b0 = ((b0 & 0xffffffu) * (m & 0xffffffu)) + b1; b1 = ((b1 & 0xffffffu) * (m & 0xffffffu)) + b2; b2 = ((b2 & 0xffffffu) * (m & 0xffffffu)) + b3; b3 = ((b3 & 0xffffffu) * (m & 0xffffffu)) + b4;
            b4 = ((b4 & 0xffffffu) * (m & 0xffffffu)) + b5; b5 = ((b5 & 0xffffffu) * (m & 0xffffffu)) + b6; b6 = ((b6 & 0xffffffu) * (m & 0xffffffu)) + b7; b7 = ((b7 & 0xffffffu) * (m & 0xffffffu)) + b0;
        }
        a0 = b0 ^ b1 ^ b2 ^ b3 ^ b4 ^ b5 ^ b6 ^ b7;
    } else {
        for (u32 i = 0; i < iters; i++) {
            a0 += a1 + m; a1 += a2 + m; a2 += a3 + m; a3 += a4 + m; a4 += a5 + m; a5 += a6 + m; a6 += a7 + m; a7 += a0 + m;
        }
    }
    u64 r = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
    if (r == 0x123456789abcdefull) sink[0] = r;
}

static inline unsigned grid_for(size_t n, unsigned bs) { return (unsigned)((n + bs - 1) / bs); }

extern "C" {
int32_t mi_gen_scalars_dev(mi_ctx *ctx, mi_fr *out_dev, size_t n, uint64_t seed, int dist) {
    if (!ctx || (!out_dev && n)) return MI_EINVAL;
    if (n) hipLaunchKernelGGL(k_gen_scalars, dim3(grid_for(n, 256)), dim3(256), 0, ctx->stream, (Fr *)out_dev, n, seed, dist);
    MI_CHECK_HIP(ctx, hipGetLastError());
    return MI_OK;
}
int32_t mi_gen_g1_dev(mi_ctx *ctx, mi_g1_affine *out_dev, size_t n, uint64_t seed) {
    if (!ctx || (!out_dev && n)) return MI_EINVAL;
    if (n) hipLaunchKernelGGL(k_gen_g1, dim3(grid_for(n, 128)), dim3(128), 0, ctx->stream, (G1Aff *)out_dev, n, seed);
    MI_CHECK_HIP(ctx, hipGetLastError());
    return MI_OK;
}
int32_t mi_gen_g2_dev(mi_ctx *ctx, mi_g2_affine *out_dev, size_t n, uint64_t seed) {
    if (!ctx || (!out_dev && n)) return MI_EINVAL;
    if (n) hipLaunchKernelGGL(k_gen_g2, dim3(grid_for(n, 64)), dim3(64), 0, ctx->stream, (G2Aff *)out_dev, n, seed);
    MI_CHECK_HIP(ctx, hipGetLastError());
    return MI_OK;
}
int32_t mi_field_op_dev(mi_ctx *ctx, int field, int op, void *z, const void *x, const void *y, size_t n) {
    if (!ctx || op < 0 || op > 8 || field < 0 || field > 1 || ((!z || !x) && n)) return MI_EINVAL;
    if (!n) return MI_OK;
    if (field == 0) hipLaunchKernelGGL(k_field_op<FrParams>, dim3(grid_for(n, 128)), dim3(128), 0, ctx->stream, op, (Fr *)z, (const Fr *)x, (const Fr *)y, n);
    else hipLaunchKernelGGL(k_field_op<FpParams>, dim3(grid_for(n, 128)), dim3(128), 0, ctx->stream, op, (Fp *)z, (const Fp *)x, (const Fp *)y, n);
    MI_CHECK_HIP(ctx, hipGetLastError());
    return MI_OK;
}
int32_t mi_g1_add_dev(mi_ctx *ctx, mi_g1_affine *out, const mi_g1_affine *a, const mi_g1_affine *b, size_t n) {
    if (!ctx || ((!out || !a || !b) && n)) return MI_EINVAL;
    if (n) hipLaunchKernelGGL(k_ec_add<Fp>, dim3(grid_for(n, 64)), dim3(64), 0, ctx->stream, (G1Aff *)out, (const G1Aff *)a, (const G1Aff *)b, n);
    MI_CHECK_HIP(ctx, hipGetLastError());
    return MI_OK;
}
int32_t mi_g2_add_dev(mi_ctx *ctx, mi_g2_affine *out, const mi_g2_affine *a, const mi_g2_affine *b, size_t n) {
    if (!ctx || ((!out || !a || !b) && n)) return MI_EINVAL;
    if (n) hipLaunchKernelGGL(k_ec_add<Fp2>, dim3(grid_for(n, 64)), dim3(64), 0, ctx->stream, (G2Aff *)out, (const G2Aff *)a, (const G2Aff *)b, n);
    MI_CHECK_HIP(ctx, hipGetLastError());
    return MI_OK;
}
int32_t mi_bench_modmul_dev(mi_ctx *ctx, int field, size_t n_threads, uint32_t iters, void *scratch_dev, float *ms_out) {
    if (!ctx || !scratch_dev || !ms_out || n_threads % 256) return MI_EINVAL;
    MI_CHECK_HIP(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
    if (field == 0) hipLaunchKernelGGL(k_bench_modmul<FrParams>, dim3((unsigned)(n_threads / 256)), dim3(256), 0, ctx->stream, (Fr *)scratch_dev, iters);
    else hipLaunchKernelGGL(k_bench_modmul<FpParams>, dim3((unsigned)(n_threads / 256)), dim3(256), 0, ctx->stream, (Fp *)scratch_dev, iters);
    MI_CHECK_HIP(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
    MI_CHECK_HIP(ctx, hipEventSynchronize(ctx->ev[1]));
    MI_CHECK_HIP(ctx, hipEventElapsedTime(ms_out, ctx->ev[0], ctx->ev[1]));
    return MI_OK;
}
// random 64-byte gathers from a table of n_entries 64-B entries (no arithmetic): the memory system's ceiling for the MSM's
// level-1 accumulation, which gathers one 64-B point per mixed addition from tables far larger than the 256 MB Infinity Cache
int32_t mi_bench_gather_dev(mi_ctx *ctx, const void *table_dev, size_t n_entries, size_t n_threads, uint32_t iters, void *scratch_dev, float *ms_out) {
    if (!ctx || !table_dev || !scratch_dev || !ms_out || n_threads % 64 || n_entries < 2 || (n_entries & (n_entries - 1))) return MI_EINVAL;
    MI_CHECK_HIP(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
    hipLaunchKernelGGL(k_bench_gather, dim3((unsigned)(n_threads / 64)), dim3(64), 0, ctx->stream, (const uint4 *)table_dev, (u64)(n_entries - 1), iters, (uint4 *)scratch_dev);
    MI_CHECK_HIP(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
    MI_CHECK_HIP(ctx, hipEventSynchronize(ctx->ev[1]));
    MI_CHECK_HIP(ctx, hipEventElapsedTime(ms_out, ctx->ev[0], ctx->ev[1]));
    return MI_OK;
}
int32_t mi_bench_valu_dev(mi_ctx *ctx, int kind, size_t n_threads, uint32_t iters, void *scratch_dev, float *ms_out) {
    if (!ctx || !scratch_dev || !ms_out || n_threads % 256) return MI_EINVAL;
    MI_CHECK_HIP(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
    hipLaunchKernelGGL(k_bench_valu, dim3((unsigned)(n_threads / 256)), dim3(256), 0, ctx->stream, kind, iters, (u64 *)scratch_dev);
    MI_CHECK_HIP(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
    MI_CHECK_HIP(ctx, hipEventSynchronize(ctx->ev[1]));
    MI_CHECK_HIP(ctx, hipEventElapsedTime(ms_out, ctx->ev[0], ctx->ev[1]));
    return MI_OK;
}
}
