// BN254 Fp / Fr in NINE 29-BIT LIMBS with Montgomery radix R' = 2^261 -- the arithmetic of the level-1 bucket accumulation.
//
// Why a second representation.  The 8 x 32-bit product of field.cuh is 136 v_mad_u64_u32 plus 120 v_addc (a 64-bit column
// accumulator overflows after one product, so every product drags a carry word along).  With 29-bit limbs a column holds up to
// eighteen products of < 2^58 -- it FITS in 64 bits -- so the product is 162 v_mad_u64_u32 and no carry instruction at all.
// Measured as dependent chains on all CUs (tools/bench_limb29): 160 G products/s against 136 G/s (+18 %).  R' = 2^261 > 64 p also
// makes the arithmetic lazy: a product of operands up to 8p x 8p comes out below 1.5p, so additions are nine limb-wise
// v_add_u32 (no carry chain, no conditional subtraction) and subtractions add a multiple of p in "borrowed" form first.
//
// Contract of the primitives (the callers in curve29.cuh track both bounds by hand; tools/f29_bounds.py re-derives them, and
// the host build of the tests traps on any violated assumption, MI_CHECK_NOWRAP):
//   limb bound L(x): every limb < 2^L.  "normalised": limbs 0..7 < 2^29 (products return this); "weak": < 2^29 + 8 (f29_wnorm)
//   value bound V(x): the number represented is < V * p
//   f29_mul(x, y)        needs L(x) + L(y) <= 60; returns a normalised value < (V(x) V(y) / 128 + 1) p
//   f29_mul2(a,b,c,d)    (ab + cd) / R' with ONE reduction; needs L + L <= 59 per pair; value < ((VaVb + VcVd) / 128 + 1) p
//   f29_mul4(a,..,h)     (ab + cd + ef + gh) / R' with ONE reduction; every operand weakly normalised; value < (sum of VV / 128 + 1) p
//   f29_add              limb-wise; L grows by one bit
//   f29_sub<K>(x, y)     x + K p - y, K p in borrowed form (limbs 0..7 in [2^30 - 2, 2^30 + 2^29)); needs y weak and V(y) < K;
//                        result limbs < 2^31
//   f29_wnorm            one independent step per limb: l_i = (l_i & M) + (l_(i-1) >> 29); weak afterwards, same value
// The data in HBM stays in gnark's 8 x 32-bit words: tables hold x * 2^261 mod p packed into 32 bytes (f29_unpack on load),
// results go back to the standard R = 2^256 form (f29_to_std) before anything else reads them.
#pragma once
#include "field.cuh"

template <class P> struct P29;
#include "field29_consts.inc"

struct F29 {
    u32 l[9];
};

#if !defined(__HIP_DEVICE_COMPILE__) && defined(MI_CHECK_NOWRAP)
#define F29_ASSERT(c) do { if (!(c)) __builtin_trap(); } while (0)
#else
#define F29_ASSERT(c) do { } while (0)
#endif

MI_HD F29 f29_zero() { F29 z;
#pragma unroll
    for (int i = 0; i < 9; i++) z.l[i] = 0;
    return z; }
template <class P>
MI_HD F29 f29_const(const u32 (&c)[9]) { F29 z;
#pragma unroll
    for (int i = 0; i < 9; i++) z.l[i] = c[i];
    return z; }

// Column accumulate.  On the device every multiply-accumulate is an OPAQUE v_mad_u64_u32 into the one column accumulator.  Left to
// itself the compiler starts a fresh accumulator per column and joins it to the shifted carry with a 64-bit add (v_lshl_add_u64:
// 17 per product at ~4.7 cycles each, against ~5 for the multiply itself -- tools/bench_valu/instr_rate.hip); the chain form measures
// 169 -> 178 G products/s at three waves per SIMD and 147 -> 169 at eight (tools/bench_valu/mul29_variants.hip,
// profiles/r02_probe_mul29_variants.txt) and 12.8 -> 13.7 G mixed additions/s in the level-1 accumulate kernel.  One asm statement per multiply (the compiler pads each with an s_nop it does not need, but stays free
// to interleave independent products; whole columns as single asm blocks measured 7 % SLOWER).  The host build (tests) is plain C
// with overflow detection under MI_CHECK_NOWRAP.
MI_HD void f29_mac(u64 &acc, u32 a, u32 b) {
#if defined(__HIP_DEVICE_COMPILE__)
    u64 carry_out;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(acc), "=s"(carry_out) : "v"(a), "v"(b));
#else
    const u64 p = (u64)a * b;
#if defined(MI_CHECK_NOWRAP)
    if (acc + p < acc) __builtin_trap();
#endif
    acc += p;
#endif
}
// the same with a wave-uniform constant as the second factor (a limb of p): it stays in a scalar register
MI_HD void f29_mac_k(u64 &acc, u32 a, u32 k) {
#if defined(__HIP_DEVICE_COMPILE__)
    u64 carry_out;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(acc), "=s"(carry_out) : "v"(a), "s"(k));
#else
    f29_mac(acc, a, k);
#endif
}

// x * y / 2^261 mod p, product scanning; no carry words (see the header)
template <class P>
MI_HD F29 f29_mul(const F29 &x, const F29 &y) {
    constexpr u32 M = (1u << 29) - 1;
    u32 m[9];
    F29 r;
    u64 acc = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) f29_mac(acc, x.l[i], y.l[k - i]);
#pragma unroll
        for (int i = 0; i < k; i++) f29_mac_k(acc, m[i], P29<P>::p[k - i]);
        m[k] = ((u32)acc * P29<P>::inv) & M;
        f29_mac_k(acc, m[k], P29<P>::p[0]);
        acc >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 17; k++) {
#pragma unroll
        for (int i = k - 8; i < 9; i++) f29_mac(acc, x.l[i], y.l[k - i]);
#pragma unroll
        for (int i = k - 8; i < 9; i++) f29_mac_k(acc, m[i], P29<P>::p[k - i]);
        r.l[k - 9] = (u32)acc & M;
        acc >>= 29;
    }
    F29_ASSERT(acc < ((u64)1 << 29));
    r.l[8] = (u32)acc;
    return r;
}
// x^2 / 2^261 mod p: 45 products x_i (2 x_j) instead of 81; same contract as f29_mul(x, x) (2 L(x) <= 60)
template <class P>
MI_HD F29 f29_sqr(const F29 &x) {
    constexpr u32 M = (1u << 29) - 1;
    u32 m[9], d[9];
    F29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) { d[i] = x.l[i] << 1; F29_ASSERT(x.l[i] < (1u << 31)); }
    u64 acc = 0;
#pragma unroll
    for (int k = 0; k < 17; k++) {
        const int lo = k < 9 ? 0 : k - 8, hi = k < 9 ? k : 8;
#pragma unroll
        for (int i = lo; i <= hi; i++) {
            const int j = k - i;
            if (i < j) f29_mac(acc, d[i], x.l[j]);
            else if (i == j) f29_mac(acc, x.l[i], x.l[i]);
        }
        if (k < 9) {
#pragma unroll
            for (int i = 0; i < k; i++) f29_mac_k(acc, m[i], P29<P>::p[k - i]);
            m[k] = ((u32)acc * P29<P>::inv) & M;
            f29_mac_k(acc, m[k], P29<P>::p[0]);
        } else {
#pragma unroll
            for (int i = k - 8; i < 9; i++) f29_mac_k(acc, m[i], P29<P>::p[k - i]);
            r.l[k - 9] = (u32)acc & M;
        }
        acc >>= 29;
    }
    F29_ASSERT(acc < ((u64)1 << 29));
    r.l[8] = (u32)acc;
    return r;
}
// (a*b + c*d) / 2^261 mod p with one reduction: 243 multiplications instead of 324
template <class P>
MI_HD F29 f29_mul2(const F29 &a, const F29 &b, const F29 &c, const F29 &d) {
    constexpr u32 M = (1u << 29) - 1;
    u32 m[9];
    F29 r;
    u64 acc = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) { f29_mac(acc, a.l[i], b.l[k - i]); f29_mac(acc, c.l[i], d.l[k - i]); }
#pragma unroll
        for (int i = 0; i < k; i++) f29_mac_k(acc, m[i], P29<P>::p[k - i]);
        m[k] = ((u32)acc * P29<P>::inv) & M;
        f29_mac_k(acc, m[k], P29<P>::p[0]);
        acc >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 17; k++) {
#pragma unroll
        for (int i = k - 8; i < 9; i++) { f29_mac(acc, a.l[i], b.l[k - i]); f29_mac(acc, c.l[i], d.l[k - i]); }
#pragma unroll
        for (int i = k - 8; i < 9; i++) f29_mac_k(acc, m[i], P29<P>::p[k - i]);
        r.l[k - 9] = (u32)acc & M;
        acc >>= 29;
    }
    F29_ASSERT(acc < ((u64)1 << 29));
    r.l[8] = (u32)acc;
    return r;
}
// (a*b + c*d + e*f + g*h) / 2^261 mod p with ONE reduction: 405 multiplications instead of 2 x 243 -- the two dual products of an Fp2
// "product minus product" (curve29_g2.cuh: Y3 = R D - Y PPP) share a reduction.  A column holds up to 36 products and 9 reduction terms:
// 45 x (2^29 + 8)^2 + the carried 2^35 < 2^63.5, so every operand must be weakly normalised (limbs < 2^29 + 8; L + L <= 58 per pair);
// value < ((VaVb + VcVd + VeVf + VgVh) / 128 + 1) p, normalised.
template <class P>
MI_HD F29 f29_mul4(const F29 &a, const F29 &b, const F29 &c, const F29 &d, const F29 &e, const F29 &f, const F29 &g, const F29 &h) {
    constexpr u32 M = (1u << 29) - 1;
    u32 m[9];
    F29 r;
    u64 acc = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) { f29_mac(acc, a.l[i], b.l[k - i]); f29_mac(acc, c.l[i], d.l[k - i]); f29_mac(acc, e.l[i], f.l[k - i]); f29_mac(acc, g.l[i], h.l[k - i]); }
#pragma unroll
        for (int i = 0; i < k; i++) f29_mac_k(acc, m[i], P29<P>::p[k - i]);
        m[k] = ((u32)acc * P29<P>::inv) & M;
        f29_mac_k(acc, m[k], P29<P>::p[0]);
        acc >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 17; k++) {
#pragma unroll
        for (int i = k - 8; i < 9; i++) { f29_mac(acc, a.l[i], b.l[k - i]); f29_mac(acc, c.l[i], d.l[k - i]); f29_mac(acc, e.l[i], f.l[k - i]); f29_mac(acc, g.l[i], h.l[k - i]); }
#pragma unroll
        for (int i = k - 8; i < 9; i++) f29_mac_k(acc, m[i], P29<P>::p[k - i]);
        r.l[k - 9] = (u32)acc & M;
        acc >>= 29;
    }
    F29_ASSERT(acc < ((u64)1 << 29));
    r.l[8] = (u32)acc;
    return r;
}
MI_HD F29 f29_add(const F29 &x, const F29 &y) {
    F29 z;
#pragma unroll
    for (int i = 0; i < 9; i++) { z.l[i] = x.l[i] + y.l[i]; F29_ASSERT(z.l[i] >= x.l[i]); }
    return z;
}
// x + K p - y with K p in borrowed form (c = P29<P>::c2 / c4 / c8): no limb can underflow for a weakly normalised y < K p
template <class P>
MI_HD F29 f29_sub(const F29 &x, const F29 &y, const u32 (&c)[9]) {
    F29 z;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        F29_ASSERT(c[i] >= y.l[i]);
        z.l[i] = x.l[i] + (c[i] - y.l[i]);
        F29_ASSERT(z.l[i] >= x.l[i]);
    }
    return z;
}
MI_HD F29 f29_wnorm(const F29 &x) {
    constexpr u32 M = (1u << 29) - 1;
    F29 z;
    z.l[0] = x.l[0] & M;
#pragma unroll
    for (int i = 1; i < 8; i++) z.l[i] = (x.l[i] & M) + (x.l[i - 1] >> 29);
    z.l[8] = x.l[8] + (x.l[7] >> 29);
    return z;
}
// exact normalisation (limbs 0..7 < 2^29) of a number whose limbs are < 2^31: one sequential carry pass, same value
MI_HD F29 f29_norm(const F29 &x) {
    constexpr u32 M = (1u << 29) - 1;
    F29 z;
    u32 carry = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { const u32 v = x.l[i] + carry; z.l[i] = v & M; carry = v >> 29; }
    z.l[8] = x.l[8] + carry;
    return z;
}
// if x >= K p (decided on the top limb: x_8 > (K p)_8, so surely x > K p): x -= K p.  kp = K p normalised.  Input weak; the result has
// limbs < 2^30 (one f29_wnorm later) and is < K p + 2^232 + (what the top-limb test cannot see) -- "almost < K p".
MI_HD F29 f29_condsub(const F29 &x, const u32 (&kp)[9]) {
    constexpr u32 B = 1u << 29;
    const bool take = x.l[8] > kp[8];
    F29 z;
    z.l[0] = x.l[0] + (take ? B - kp[0] : 0u);
#pragma unroll
    for (int i = 1; i < 8; i++) z.l[i] = x.l[i] + (take ? B - 1 - kp[i] : 0u);
    z.l[8] = x.l[8] - (take ? kp[8] + 1 : 0u);
    return z;
}
// 32 bytes (8 x u32 little-endian, any value < 2^256) -> nine 29-bit limbs, normalised
MI_HD F29 f29_unpack(const u32 *w) {
    constexpr u32 M = (1u << 29) - 1;
    F29 z;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const int bit = 29 * i, k = bit >> 5, sh = bit & 31;
        u32 v = w[k] >> sh;
        if (sh > 3 && k + 1 < 8) v |= w[k + 1] << (32 - sh);   // sh > 3: the limb straddles two words
        z.l[i] = i < 8 ? (v & M) : v;
    }
    return z;
}
// exact value of a NORMALISED number < 2^256, packed back into 8 x u32
MI_HD void f29_pack(const F29 &x, u32 *w) {
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const int bit = 32 * k, i = bit / 29, sh = bit - 29 * i;   // word k starts inside limb i
        u64 v = (u64)x.l[i] >> sh;
        int have = 29 - sh;
        for (int j = i + 1; have < 32 && j < 9; j++) { v |= (u64)x.l[j] << have; have += 29; }
        w[k] = (u32)v;
    }
}
// value of x in the standard R = 2^256 Montgomery form, canonical: mul'(x, 2^256) = x_plain * 2^256, then one exact reduction
template <class P>
MI_HD Fe<P> f29_to_std(const F29 &x) {
    constexpr u32 M = (1u << 29) - 1;
    F29 t = f29_mul<P>(x, f29_const<P>(P29<P>::to_std));   // normalised, < 2p for V(x) <= 128
    // t >= p ?  subtract p with a borrow chain over the 29-bit limbs and keep the difference if it did not go negative
    F29 d;
    u32 borrow = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        u32 v = t.l[i] - P29<P>::p[i] - borrow;
        borrow = v >> 31;              // limbs are < 2^29 (top < 2^24): a negative difference has bit 31 set
        d.l[i] = i < 8 ? (v & M) : v;
    }
    if (!borrow) t = d;
    Fe<P> r;
    f29_pack(t, r.l);
    return r;
}
// standard-form canonical element -> R' form (normalised, < 1.01 p)
template <class P>
MI_HD F29 f29_from_std(const Fe<P> &x) { return f29_mul<P>(f29_unpack(x.l), f29_const<P>(P29<P>::from_std)); }
// x * 2^5 mod p in the STANDARD arithmetic: the 32 bytes a table keeps for the R' form of a standard-form coordinate
template <class P>
MI_HD Fe<P> fe_to_rprime_packed(const Fe<P> &x) {
    Fe<P> t = x + x; t = t + t; t = t + t; t = t + t; t = t + t;
    return t;
}
