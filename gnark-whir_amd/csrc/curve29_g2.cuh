// G2 bucket accumulation in the 9 x 29-bit representation (field29.cuh): Fp2 = Fp[u]/(u^2 + 1) over lazy 29-bit-limb Fp, and the
// XYZZ mixed addition with hand-tracked bounds.  Same group law and special cases as xyzz_madd (curve.cuh); replaces it inside
// the G2 level-1 accumulate kernel, and (g2x29_add, at the end of this file) xyzz_add in the levels above it.
//
// An Fp2 product is two DUAL products (field29.cuh f29_mul2: two multiplications accumulated in the same 64-bit columns, one
// Montgomery reduction): c0 = a0 b0 + a1 (K p - b1), c1 = a0 b1 + a1 b0 -- 486 multiplications, no Karatsuba additions; a
// square is (a0 + a1)(a0 + K p - a1) and a0 (2 a1): 324.  The 8 x 32-bit form spends 3 x (136 + 120 carry additions) per product.
//
// Bounds (V in multiples of p; every operand of a product is weakly normalised, limbs < 2^29 + 8):
//   accumulator   X < 2.1 (reduced below), Y < 3.6, ZZ, ZZZ < 1.1                      point (table)  x, y canonical
//   U2 = ZZ x2  < 1.04      S2 = ZZZ y2 < 1.04      P = U2 + 4p - X  < 5.1      R = S2 + 4p - Y  < 5.1
//   PP = P^2: (2 * 5.1)(5.1 + 8) / 128 + 1 = 2.05, 5.1 * 10.2 / 128 + 1 = 1.41      RR = R^2: the same
//   PPP = P PP   < 1.25     Q = X PP   < 1.11      T = PPP + 2Q  < 3.5      X3 = RR + 4p - T  < 6.1  -> conditional -4p, -2p: < 2.1
//   D = Q + 4p - X3  < 5.3     Y3 = (R D - Y PPP) / R' as ONE reduction per component (f2_29_mul_sub): (5.1 * 5.3 + 5.1 * 8 + 3.6 * 2 + 3.6 * 1.25) / 128 + 1 < 1.7
//   (r6; before: M1 = R D, M2 = Y PPP, Y3 = M1 + 2p - M2 < 3.6 -- the accumulator's Y bound stays the looser 3.6 below)
//   ZZ3 = ZZ PP < 1.05      ZZZ3 = ZZZ PPP < 1.03
#pragma once
#include "curve.cuh"
#include "field29.cuh"

struct F2_29 {
    F29 a0, a1;
};
typedef FpParams P_;
MI_HD F2_29 f2_29_wnorm(const F2_29 &x) { return F2_29{f29_wnorm(x.a0), f29_wnorm(x.a1)}; }
MI_HD F2_29 f2_29_add(const F2_29 &x, const F2_29 &y) { return F2_29{f29_add(x.a0, y.a0), f29_add(x.a1, y.a1)}; }
// x + K p - y per component (K p borrowed: c2 / c4 / c8), weakly normalised afterwards
MI_HD F2_29 f2_29_sub(const F2_29 &x, const F2_29 &y, const u32 (&c)[9]) {
    return F2_29{f29_wnorm(f29_sub<P_>(x.a0, y.a0, c)), f29_wnorm(f29_sub<P_>(x.a1, y.a1, c))};
}
// x * y; the offset ck = K p (borrowed) must exceed y.a1
MI_HD F2_29 f2_29_mul(const F2_29 &x, const F2_29 &y, const u32 (&ck)[9]) {
    const F29 n1 = f29_wnorm(f29_sub<P_>(f29_zero(), y.a1, ck));
    return F2_29{f29_mul2<P_>(x.a0, y.a0, x.a1, n1), f29_mul2<P_>(x.a0, y.a1, x.a1, y.a0)};
}
// x^2 for V(x) < 8
MI_HD F2_29 f2_29_sqr(const F2_29 &x) {
    const F29 s = f29_wnorm(f29_add(x.a0, x.a1)), d = f29_wnorm(f29_sub<P_>(x.a0, x.a1, P29<P_>::c8));
    return F2_29{f29_mul<P_>(s, d), f29_mul<P_>(x.a0, f29_add(x.a1, x.a1))};
}
// x * y - z * w with ONE reduction per component (f29_mul4): a0 = x0 y0 + x1 (Ky p - y1) + z0 (Kw p - w0) + z1 w1,
// a1 = x0 y1 + x1 y0 + z0 (Kw p - w1) + z1 (Kw p - w0).  cy = Ky p, cw = Kw p (borrowed) must exceed V(y), V(w); every operand weakly
// normalised.  810 multiplications against the 972 of two products and a subtraction; the result is normalised.
MI_HD F2_29 f2_29_mul_sub(const F2_29 &x, const F2_29 &y, const u32 (&cy)[9], const F2_29 &z, const F2_29 &w, const u32 (&cw)[9]) {
    const F29 ny1 = f29_wnorm(f29_sub<P_>(f29_zero(), y.a1, cy));
    const F29 nw0 = f29_wnorm(f29_sub<P_>(f29_zero(), w.a0, cw)), nw1 = f29_wnorm(f29_sub<P_>(f29_zero(), w.a1, cw));
    return F2_29{f29_mul4<P_>(x.a0, y.a0, x.a1, ny1, z.a0, nw0, z.a1, w.a1), f29_mul4<P_>(x.a0, y.a1, x.a1, y.a0, z.a0, nw1, z.a1, nw0)};
}
// an "almost < 2p" representative of a value < 6.1 p
MI_HD F29 f29_below_2p(const F29 &x) { return f29_wnorm(f29_condsub(f29_wnorm(f29_condsub(x, P29<P_>::p4)), P29<P_>::p2)); }
// 16 packed words (a0 | a1, each the canonical R' value in 8 x u32) <-> F2_29
MI_HD F2_29 f2_29_unpack(const u32 *w) { return F2_29{f29_unpack(w), f29_unpack(w + 8)}; }
MI_HD Fp2 f2_29_to_std(const F2_29 &x) { return Fp2{f29_to_std<P_>(x.a0), f29_to_std<P_>(x.a1)}; }
MI_HD F2_29 f2_29_from_std(const Fp2 &x) { return F2_29{f29_from_std<P_>(x.a0), f29_from_std<P_>(x.a1)}; }
MI_HD bool f2_29_is_zero_mod_p(const F2_29 &x) {   // x = a normalised product output < 2.1 p per component: zero iff each is 0, p or 2p
    bool z = true;
    const F29 *c[2] = {&x.a0, &x.a1};
    for (int k = 0; k < 2; k++) {
        u32 e0 = 0, e1 = 0, e2 = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) { e0 |= c[k]->l[i]; e1 |= c[k]->l[i] ^ P29<P_>::p[i]; e2 |= c[k]->l[i] ^ P29<P_>::p2[i]; }
        z = z && (!e0 || !e1 || !e2);
    }
    return z;
}

// Acc: ld(comp) / st(comp, F2_29) for comp = 0 (X), 1 (Y), 2 (ZZ), 3 (ZZZ) -- registers on the host, an LDS image on the device.
// q: 32 packed words x.a0 | x.a1 | y.a0 | y.a1 in the R' form; all zero = infinity.  inf: "the accumulator is the point at infinity".
template <class Acc>
MI_HD void g2x29_madd(Acc &A, bool &inf, const u32 *q, bool negate) {
    u32 any = 0;
#pragma unroll
    for (int i = 0; i < 32; i++) any |= q[i];
    if (!any) return;
    const F2_29 x2 = f2_29_unpack(q);
    F2_29 y2;
    if (negate) {   // p - y per component on the packed canonical words; a zero component stays zero
        Fp t, n;
        u32 w[16];
        for (int k = 0; k < 2; k++) {
#pragma unroll
            for (int i = 0; i < 8; i++) t.l[i] = q[16 + 8 * k + i];
            n = fe_neg(t);
#pragma unroll
            for (int i = 0; i < 8; i++) w[8 * k + i] = n.l[i];
        }
        y2 = f2_29_unpack(w);
    } else {
        y2 = f2_29_unpack(q + 16);
    }
    if (inf) {
        const F29 one = f29_const<P_>(P29<P_>::one);
        A.st(0, x2); A.st(1, y2); A.st(2, F2_29{one, f29_zero()}); A.st(3, F2_29{one, f29_zero()});
        inf = false;
        return;
    }
    const F2_29 U2 = f2_29_mul(A.ld(2), x2, P29<P_>::c2);
    const F2_29 S2 = f2_29_mul(A.ld(3), y2, P29<P_>::c2);
    const F2_29 X = A.ld(0);
    const F2_29 Pp = f2_29_sub(U2, X, P29<P_>::c4);
    const F2_29 PP = f2_29_sqr(Pp);
    if (f2_29_is_zero_mod_p(PP)) {   // P = 0 (P^2 = 0 in a field): doubling or cancellation -- rare: the standard arithmetic handles it
        G2X s{f2_29_to_std(X), f2_29_to_std(A.ld(1)), f2_29_to_std(A.ld(2)), f2_29_to_std(A.ld(3))};
        G2Aff qs{f2_29_to_std(x2), f2_29_to_std(y2)};
        xyzz_madd(s, qs, false);
        inf = s.is_inf();
        if (!inf) { A.st(0, f2_29_from_std(s.x)); A.st(1, f2_29_from_std(s.y)); A.st(2, f2_29_from_std(s.zz)); A.st(3, f2_29_from_std(s.zzz)); }
        return;
    }
    const F2_29 PPP = f2_29_mul(Pp, PP, P29<P_>::c4);
    const F2_29 Q = f2_29_mul(X, PP, P29<P_>::c4);
    A.st(2, f2_29_mul(A.ld(2), PP, P29<P_>::c4));
    A.st(3, f2_29_mul(A.ld(3), PPP, P29<P_>::c2));
    const F2_29 Y = A.ld(1);
    const F2_29 R = f2_29_sub(S2, Y, P29<P_>::c4);
    const F2_29 RR = f2_29_sqr(R);
    const F2_29 T = f2_29_wnorm(f2_29_add(f2_29_add(PPP, Q), Q));
    F2_29 X3 = f2_29_sub(RR, T, P29<P_>::c4);
    X3.a0 = f29_below_2p(X3.a0); X3.a1 = f29_below_2p(X3.a1);
    A.st(0, X3);
    const F2_29 D = f2_29_sub(Q, X3, P29<P_>::c4);
    A.st(1, f2_29_mul_sub(R, D, P29<P_>::c8, Y, PPP, P29<P_>::c2));   // Y3 = R D - Y PPP: one reduction per component
}

// ---- partial sums in the packed R' form (the G2 twin of g1x29_store_rp / g1x29_load_rp / g1x29_add in curve29.cuh)
// An accumulator component -> 16 words (a0 | a1); every stored coordinate is below 3.6 p < 2^256 (bounds in the header).
MI_HD void f2_29_pack(const F2_29 &x, u32 *w) { f29_pack(f29_norm(x.a0), w); f29_pack(f29_norm(x.a1), w + 8); }
// Acc += b, b = 64 packed words X | Y | ZZ | ZZZ read through ldb(comp) (the caller decides where they live: global memory, read when
// needed, so that only the operands of the current step occupy registers).  add-2008-s, same special cases as xyzz_add (curve.cuh).
// Bounds: both operands X < 2.1, Y < 3.6, ZZ, ZZZ < 1.1 (what g2x29_madd and this function leave behind)
//   U1 = Xa ZZb, U2 = Xb ZZa, S1 = Ya ZZZb, S2 = Yb ZZZa < 1.1      P = U2 + 2p - U1, R = S2 + 2p - S1 < 3.1
//   PP = P^2 < 1.6 / 1.2     PPP = P PP < 1.1     Q = U1 PP < 1.1     T = PPP + 2Q < 3.3     X3 = R^2 + 4p - T < 5.6 -> below 2p: < 2.1
//   D = Q + 4p - X3 < 5.1    Y3 = (R D - S1 PPP) / R', one reduction per component: (3.1 * 5.1 + 3.1 * 8 + 1.1 * 2 + 1.1 * 1.1) / 128 + 1 < 1.4
//   ZZ3 = (ZZa ZZb) PP < 1.1     ZZZ3 = (ZZZa ZZZb) PPP < 1.1
template <class Acc, class LoadB>
MI_HD void g2x29_add(Acc &A, bool &inf, const LoadB &ldb, bool b_inf) {
    if (b_inf) return;
    if (inf) {
        A.st(0, ldb(0)); A.st(1, ldb(1)); A.st(2, ldb(2)); A.st(3, ldb(3));
        inf = false;
        return;
    }
    const F2_29 ZZb = ldb(2);
    const F2_29 U1 = f2_29_mul(A.ld(0), ZZb, P29<P_>::c2);
    const F2_29 Pp = f2_29_sub(f2_29_mul(ldb(0), A.ld(2), P29<P_>::c2), U1, P29<P_>::c2);
    const F2_29 PP = f2_29_sqr(Pp);
    if (f2_29_is_zero_mod_p(PP)) {   // equal x: doubling or cancellation -- rare: the standard arithmetic handles it
        G2X sa{f2_29_to_std(A.ld(0)), f2_29_to_std(A.ld(1)), f2_29_to_std(A.ld(2)), f2_29_to_std(A.ld(3))};
        const G2X sb{f2_29_to_std(ldb(0)), f2_29_to_std(ldb(1)), f2_29_to_std(ZZb), f2_29_to_std(ldb(3))};
        xyzz_add(sa, sb);
        inf = sa.is_inf();
        if (!inf) { A.st(0, f2_29_from_std(sa.x)); A.st(1, f2_29_from_std(sa.y)); A.st(2, f2_29_from_std(sa.zz)); A.st(3, f2_29_from_std(sa.zzz)); }
        return;
    }
    A.st(2, f2_29_mul(f2_29_mul(A.ld(2), ZZb, P29<P_>::c2), PP, P29<P_>::c2));
    const F2_29 PPP = f2_29_mul(Pp, PP, P29<P_>::c2);
    const F2_29 Q = f2_29_mul(U1, PP, P29<P_>::c2);
    const F2_29 ZZZb = ldb(3);
    const F2_29 S1 = f2_29_mul(A.ld(1), ZZZb, P29<P_>::c2);
    const F2_29 R = f2_29_sub(f2_29_mul(ldb(1), A.ld(3), P29<P_>::c2), S1, P29<P_>::c2);
    A.st(3, f2_29_mul(f2_29_mul(A.ld(3), ZZZb, P29<P_>::c2), PPP, P29<P_>::c2));
    const F2_29 RR = f2_29_sqr(R);
    const F2_29 T = f2_29_wnorm(f2_29_add(f2_29_add(PPP, Q), Q));
    F2_29 X3 = f2_29_sub(RR, T, P29<P_>::c4);
    X3.a0 = f29_below_2p(X3.a0); X3.a1 = f29_below_2p(X3.a1);
    A.st(0, X3);
    const F2_29 D = f2_29_sub(Q, X3, P29<P_>::c4);
    A.st(1, f2_29_mul_sub(R, D, P29<P_>::c8, S1, PPP, P29<P_>::c2));   // Y3 = R D - S1 PPP
}
