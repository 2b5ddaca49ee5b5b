// BN254 G1 (over Fp) and G2 (twist over Fp2) group law for the MSM kernels.
// Replaces gnark-crypto ecc/bn254 g1.go / g2.go point arithmetic used by MultiExp on the path
// reached from /root/reference/mt.go:496.
//
// Buckets and partial sums are kept in extended-Jacobian "XYZZ" coordinates
// (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2): mixed addition of an affine pk point is 8M+2S with
// no inversion; infinity is ZZ == 0.  Affine infinity is (0,0) as gnark encodes it.
#pragma once
#include "field.cuh"

template <class F>
struct Affine {
    F x, y;
    MI_HD bool is_inf() const { return x.is_zero() && y.is_zero(); }
};
template <class F>
struct XYZZ {
    F x, y, zz, zzz;
    MI_HD bool is_inf() const { return zz.is_zero(); }
    static MI_HD XYZZ inf() { return XYZZ{F::one(), F::one(), F::zero(), F::zero()}; }
    static MI_HD XYZZ from_affine(const Affine<F> &p) {
        if (p.is_inf()) return inf();
        return XYZZ{p.x, p.y, F::one(), F::one()};
    }
};
template <class F>
struct Jac {
    F x, y, z;
};

typedef Affine<Fp> G1Aff;
typedef Affine<Fp2> G2Aff;
typedef XYZZ<Fp> G1X;
typedef XYZZ<Fp2> G2X;

// dbl-2008-s-1 (a = 0): works on XYZZ input
template <class F>
MI_HD XYZZ<F> xyzz_dbl(const XYZZ<F> &p) {
    if (p.is_inf()) return p;
    F U = fe_dbl(p.y);
    F V = fe_sqr(U);
    F W = U * V;
    F S = p.x * V;
    F X2 = fe_sqr(p.x);
    F M = fe_dbl(X2) + X2;
    F X3 = fe_sqr(M) - fe_dbl(S);
    F Y3 = fe_mul_sub(M, S - X3, W, p.y);
    return XYZZ<F>{X3, Y3, V * p.zz, W * p.zzz};
}
// doubling of an affine point straight to XYZZ (mdbl-2008-s-1)
template <class F>
MI_HD XYZZ<F> xyzz_dbl_affine(const F &x, const F &y) {
    F U = fe_dbl(y);
    F V = fe_sqr(U);
    F W = U * V;
    F S = x * V;
    F X2 = fe_sqr(x);
    F M = fe_dbl(X2) + X2;
    F X3 = fe_sqr(M) - fe_dbl(S);
    F Y3 = fe_mul_sub(M, S - X3, W, y);
    return XYZZ<F>{X3, Y3, V, W};
}
// acc += (+/-) q, q affine (madd-2008-s); handles inf / equal / opposite operands
template <class F>
MI_HD void xyzz_madd(XYZZ<F> &acc, const Affine<F> &q, bool negate) {
    if (q.is_inf()) return;
    F qy = negate ? fe_neg(q.y) : q.y;
    if (acc.is_inf()) {
        acc = XYZZ<F>{q.x, qy, F::one(), F::one()};
        return;
    }
    F U2 = q.x * acc.zz;
    F S2 = qy * acc.zzz;
    F Pp = U2 - acc.x;
    F R = S2 - acc.y;
    if (Pp.is_zero()) {
        if (R.is_zero()) acc = xyzz_dbl_affine(q.x, qy);
        else acc = XYZZ<F>::inf();
        return;
    }
    F PP = fe_sqr(Pp);
    F PPP = Pp * PP;
    F Q = acc.x * PP;
    F X3 = fe_sqr(R) - PPP - fe_dbl(Q);
    F Y3 = fe_mul_sub(R, Q - X3, acc.y, PPP);   // two products, one Montgomery reduction (field.cuh)
    acc = XYZZ<F>{X3, Y3, acc.zz * PP, acc.zzz * PPP};
}
// acc += q, both XYZZ (add-2008-s)
template <class F>
MI_HD void xyzz_add(XYZZ<F> &acc, const XYZZ<F> &q) {
    if (q.is_inf()) return;
    if (acc.is_inf()) { acc = q; return; }
    F U1 = acc.x * q.zz;
    F U2 = q.x * acc.zz;
    F S1 = acc.y * q.zzz;
    F S2 = q.y * acc.zzz;
    F Pp = U2 - U1;
    F R = S2 - S1;
    if (Pp.is_zero()) {
        if (R.is_zero()) acc = xyzz_dbl(acc);
        else acc = XYZZ<F>::inf();
        return;
    }
    F PP = fe_sqr(Pp);
    F PPP = Pp * PP;
    F Q = U1 * PP;
    F X3 = fe_sqr(R) - PPP - fe_dbl(Q);
    F Y3 = fe_mul_sub(R, Q - X3, S1, PPP);
    acc = XYZZ<F>{X3, Y3, acc.zz * q.zz * PP, acc.zzz * q.zzz * PPP};
}
template <class F>
MI_HD XYZZ<F> xyzz_neg(const XYZZ<F> &p) { return XYZZ<F>{p.x, fe_neg(p.y), p.zz, p.zzz}; }

template <class F>
MI_HD Affine<F> xyzz_to_affine(const XYZZ<F> &p) {
    if (p.is_inf()) return Affine<F>{F::zero(), F::zero()};
    // 1/ZZZ then 1/ZZ = ZZZ^-2 * ZZ^2  (ZZ^3 = ZZZ^2  =>  1/ZZ = ZZ^2 / ZZZ^2)
    F izzz = fe_inv(p.zzz);
    F izz = fe_sqr(izzz) * fe_sqr(p.zz);
    return Affine<F>{p.x * izz, p.y * izzz};
}
// k * p for a small plain integer k (bucket-reduce segment offsets), double-and-add
template <class F>
MI_HD XYZZ<F> xyzz_mul_u32(const XYZZ<F> &p, u32 k) {
    XYZZ<F> acc = XYZZ<F>::inf();
    for (int i = 31; i >= 0; i--) {
        acc = xyzz_dbl(acc);
        if ((k >> i) & 1) xyzz_add(acc, p);
    }
    return acc;
}
// k * p, k = 8 x u32 plain integer
template <class F>
MI_HD XYZZ<F> xyzz_mul_256(const XYZZ<F> &p, const u32 k[8]) {
    XYZZ<F> acc = XYZZ<F>::inf();
    for (int i = 255; i >= 0; i--) {
        acc = xyzz_dbl(acc);
        if ((k[i >> 5] >> (i & 31)) & 1) xyzz_add(acc, p);
    }
    return acc;
}
// curve constant b: G1 b = 3; G2 b' = 3/(9+u)
MI_HD Fp curve_b(const Fp *) { return fe_from_u32<FpParams>(3); }
MI_HD Fp2 curve_b(const Fp2 *) {
    Fp2 nine_u{fe_from_u32<FpParams>(9), Fp::one()};
    Fp2 three{fe_from_u32<FpParams>(3), Fp::zero()};
    return three * fe_inv(nine_u);
}
