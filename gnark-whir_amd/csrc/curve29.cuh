// G1 bucket accumulation in the 9 x 29-bit representation (field29.cuh): the XYZZ mixed addition with hand-tracked lazy bounds.
// Same group law and the same special cases as xyzz_madd (curve.cuh); replaces it inside the level-1 accumulate kernel only.
//
// Bounds (V = value bound in multiples of p, see field29.cuh; tools/f29_bounds.py replays this table):
//   accumulator invariant   X weak, V < 5.7 | Y normalised, V < 1.8 (V <= 1 right after the first point) | ZZ, ZZZ normalised, V < 1.04
//   U2 = x2 ZZ      1.01      S2 = y2 ZZZ     1.01
//   P  = U2 + 8p - X   < 9.1  (weak)           R  = S2 + 8p - Y   < 9.1  (weak)
//   PP = P^2        1.65      PPP = P PP      1.12      Q = X PP      1.08      RR = R^2   1.65
//   T  = PPP + 2Q   < 3.3 (weak)               X3 = RR + 4p - T   < 5.7 (weak)
//   D  = Q + 8p - X3   < 9.1 (weak)            nY = 8p - Y        <= 8  (weak)
//   Y3 = (R D + nY PPP) / R'   < (9.1 * 9.1 + 8 * 1.12) / 128 + 1 = 1.72
//   ZZ3 = ZZ PP     1.02      ZZZ3 = ZZZ PPP  1.01
//
// Partial sums between the levels of the item machinery stay in the R' form ("rp": four coordinates packed into 8 x u32 each, NOT
// canonical -- any representative below 2^256 ~ 5.29 p): the item's end then costs a pack instead of four conversion products, and
// the next level adds two such sums with g1x29_add (add-2008-s, 12 products + 2 squares, one dual product) without converting:
//   a = running sum: X weak V < 5.7 | Y < 1.8 | ZZ, ZZZ < 1.04        b = loaded partial: X < 4.02 | Y < 1.8 | ZZ, ZZZ < 1.04 (normalised)
//   U1 = Xa ZZb 1.05   U2 = Xb ZZa 1.04   S1 = Ya ZZZb 1.02   S2 = Yb ZZZa 1.02   P = U2 + 2p - U1 < 3.04   R = S2 + 2p - S1 < 3.02
//   PP = P^2 1.08   PPP = P PP 1.03   Q = U1 PP 1.01   T = PPP + 2Q < 3.05   RR = R^2 1.08   X3 = RR + 4p - T < 5.07 (weak)
//   D = Q + 8p - X3 < 9.01   nS1 = 2p - S1 <= 2   Y3 = (R D + nS1 PPP) / R' < 1.23   ZZ3 = ZZa ZZb PP 1.01   ZZZ3 = ZZZa ZZZb PPP 1.01
#pragma once
#include "curve.cuh"
#include "field29.cuh"

struct G1X29 {
    F29 x, y, zz, zzz;
    bool inf;
};
MI_HD G1X29 g1x29_inf() { G1X29 a; a.x = a.y = a.zz = a.zzz = f29_zero(); a.inf = true; return a; }

// accumulator -> the standard XYZZ the rest of the MSM works with
MI_HD G1X g1x29_to_std(const G1X29 &a) {
    if (a.inf) return G1X::inf();
    return G1X{f29_to_std<FpParams>(a.x), f29_to_std<FpParams>(a.y), f29_to_std<FpParams>(a.zz), f29_to_std<FpParams>(a.zzz)};
}
MI_HD G1X29 g1x29_from_std(const G1X &s) {
    if (s.is_inf()) return g1x29_inf();
    G1X29 a;
    a.x = f29_from_std<FpParams>(s.x); a.y = f29_from_std<FpParams>(s.y); a.zz = f29_from_std<FpParams>(s.zz); a.zzz = f29_from_std<FpParams>(s.zzz);
    a.inf = false;
    return a;
}

// acc += (+/-) q.  q = 16 words: x | y, each the canonical value of coordinate * 2^261 mod p packed in 8 x u32 (what the
// tables of the fixed-base path / the converted bases hold); (0, 0) = infinity.
MI_HD void g1x29_madd(G1X29 &acc, const u32 *q, bool negate) {
    typedef FpParams P;
    u32 any = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) any |= q[i];
    if (!any) return;
    const F29 x2 = f29_unpack(q);
    F29 y2;
    if (negate) {   // p - y on the packed canonical words (exact), then unpack
        Fp yy, ny;
#pragma unroll
        for (int i = 0; i < 8; i++) yy.l[i] = q[8 + i];
        fe_sub_raw(ny, Fp::modulus(), yy);
        y2 = f29_unpack(ny.l);
    } else {
        y2 = f29_unpack(q + 8);
    }
    if (acc.inf) {
        acc.x = x2; acc.y = y2; acc.zz = f29_const<P>(P29<P>::one); acc.zzz = acc.zz; acc.inf = false;
        return;
    }
    const F29 U2 = f29_mul<P>(x2, acc.zz);
    const F29 S2 = f29_mul<P>(y2, acc.zzz);
    const F29 Pp = f29_wnorm(f29_sub<P>(U2, acc.x, P29<P>::c8));
    const F29 R = f29_wnorm(f29_sub<P>(S2, acc.y, P29<P>::c8));
    const F29 PP = f29_sqr<P>(Pp);
    // P = 0 mod p (same x: doubling or cancellation)  <=>  PP in {0, p}; PP is normalised, so its limbs decide.  Rare: the
    // standard arithmetic handles it.
    if (PP.l[0] == 0 || PP.l[0] == P29<P>::p[0]) {
        u32 z = 0, e = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) { z |= PP.l[i]; e |= PP.l[i] ^ P29<P>::p[i]; }
        if (!z || !e) {
            G1X s = g1x29_to_std(acc);
            G1Aff qs{f29_to_std<P>(x2), f29_to_std<P>(y2)};
            xyzz_madd(s, qs, false);
            acc = g1x29_from_std(s);
            return;
        }
    }
    const F29 PPP = f29_mul<P>(Pp, PP);
    const F29 Q = f29_mul<P>(acc.x, PP);
    const F29 T = f29_wnorm(f29_add(f29_add(PPP, Q), Q));
    const F29 RR = f29_sqr<P>(R);
    const F29 X3 = f29_wnorm(f29_sub<P>(RR, T, P29<P>::c4));
    const F29 D = f29_wnorm(f29_sub<P>(Q, X3, P29<P>::c8));
    const F29 nY = f29_wnorm(f29_sub<P>(f29_zero(), acc.y, P29<P>::c8));
    acc.y = f29_mul2<P>(R, D, nY, PPP);
    acc.x = X3;
    acc.zz = f29_mul<P>(acc.zz, PP);
    acc.zzz = f29_mul<P>(acc.zzz, PPP);
}

// ---- partial sums in the packed R' form
// acc -> 32 words X | Y | ZZ | ZZZ.  X (weak, < 5.7 p) is brought below 4p + 2^233 < 2^256 first; the others are products' results
// (normalised, < 2 p).  Infinity = all zero, like XYZZ::inf().
MI_HD void g1x29_store_rp(const G1X29 &a, u32 *w) {
    if (a.inf) {
#pragma unroll
        for (int i = 0; i < 32; i++) w[i] = 0;
        return;
    }
    const F29 x = f29_norm(f29_condsub(a.x, P29<FpParams>::p4));
    F29_ASSERT(x.l[8] < (1u << 24));
    f29_pack(x, w); f29_pack(a.y, w + 8); f29_pack(a.zz, w + 16); f29_pack(a.zzz, w + 24);
}
MI_HD G1X29 g1x29_load_rp(const u32 *w) {
    G1X29 a;
    u32 any = 0;
#pragma unroll
    for (int i = 16; i < 24; i++) any |= w[i];
    a.inf = any == 0;   // ZZ = 0 exactly: only the stored infinity (a finite point's ZZ is not 0 mod p, and 0 mod p is stored as 0, p or 2p only for it)
    a.x = f29_unpack(w); a.y = f29_unpack(w + 8); a.zz = f29_unpack(w + 16); a.zzz = f29_unpack(w + 24);
    return a;
}
// a += b, both XYZZ in the R' form (bounds in the header); same special cases as xyzz_add (curve.cuh)
MI_HD void g1x29_add(G1X29 &a, const G1X29 &b) {
    typedef FpParams P;
    if (b.inf) return;
    if (a.inf) { a = b; return; }
    const F29 U1 = f29_mul<P>(a.x, b.zz), U2 = f29_mul<P>(b.x, a.zz);
    const F29 S1 = f29_mul<P>(a.y, b.zzz), S2 = f29_mul<P>(b.y, a.zzz);
    const F29 Pp = f29_wnorm(f29_sub<P>(U2, U1, P29<P>::c2));
    const F29 R = f29_wnorm(f29_sub<P>(S2, S1, P29<P>::c2));
    const F29 PP = f29_sqr<P>(Pp);
    if (PP.l[0] == 0 || PP.l[0] == P29<P>::p[0]) {   // P = 0 mod p (equal x): doubling or cancellation -- rare, in the standard arithmetic
        u32 z = 0, e = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) { z |= PP.l[i]; e |= PP.l[i] ^ P29<P>::p[i]; }
        if (!z || !e) {
            G1X sa = g1x29_to_std(a);
            xyzz_add(sa, g1x29_to_std(b));
            a = g1x29_from_std(sa);
            return;
        }
    }
    const F29 PPP = f29_mul<P>(Pp, PP);
    const F29 Q = f29_mul<P>(U1, PP);
    const F29 T = f29_wnorm(f29_add(f29_add(PPP, Q), Q));
    const F29 RR = f29_sqr<P>(R);
    const F29 X3 = f29_wnorm(f29_sub<P>(RR, T, P29<P>::c4));
    const F29 D = f29_wnorm(f29_sub<P>(Q, X3, P29<P>::c8));
    const F29 nS1 = f29_wnorm(f29_sub<P>(f29_zero(), S1, P29<P>::c2));
    a.y = f29_mul2<P>(R, D, nS1, PPP);
    a.x = X3;
    a.zz = f29_mul<P>(f29_mul<P>(a.zz, b.zz), PP);
    a.zzz = f29_mul<P>(f29_mul<P>(a.zzz, b.zzz), PPP);
}
