/* CPU restatement, field layer: 4 x u64 little-endian limbs, Montgomery form, R = 2^256.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/pyref.py header).  PARITY UNPINNED: the reference's
 * arithmetic for this path is gnark-crypto v0.14.1-0.20241217131346-b998989abdbe
 * ecc/bn254/{fr,fp} (go.mod:7), absent from the container; this file restates textbook CIOS
 * Montgomery multiplication.  Limb order and the Fr modulus follow
 * /root/reference/typeConverters/typeConverters.go:26-44.  Checked against oracle/pyref.py
 * (Python big integers) in tests/test_oracle.py.
 */
#ifndef REF_FIELD_H
#define REF_FIELD_H
#include <stdint.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef struct { uint64_t l[4]; } fe;

typedef struct {
    fe p;        /* modulus */
    fe r2;       /* R^2 mod p */
    fe one;      /* R mod p */
    uint64_t inv; /* -p^-1 mod 2^64 */
} fctx;

/* constants VERIFIED in SURVEY.md section 8a row a11; re-derived in tests/test_oracle.py */
static const fctx FR = {
    {{0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL}},
    {{0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL, 0x8c49833d53bb8085ULL, 0x0216d0b17f4e44a5ULL}},
    {{0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL, 0x0e0a77c19a07df2fULL}},
    0xc2e1f593efffffffULL};
static const fctx FP = {
    {{0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL}},
    {{0xf32cfc5b538afa89ULL, 0xb5e71911d44501fbULL, 0x47ab1eff0a417ff6ULL, 0x06d89f71cab8351fULL}},
    {{0xd35d438dc58f0d9dULL, 0x0a78eb28f5c70b3dULL, 0x666ea36f7879462cULL, 0x0e0a77c19a07df2fULL}},
    0x87d20782e4866389ULL};

static inline int fe_is_zero(const fe *a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3]) == 0; }
static inline int fe_eq(const fe *a, const fe *b) { return memcmp(a, b, sizeof(fe)) == 0; }
static inline int fe_geq(const fe *a, const fe *b) {
    for (int i = 3; i >= 0; i--) {
        if (a->l[i] > b->l[i]) return 1;
        if (a->l[i] < b->l[i]) return 0;
    }
    return 1;
}
static inline uint64_t fe_add_raw(fe *z, const fe *x, const fe *y) {
    u128 c = 0;
    for (int i = 0; i < 4; i++) { c += (u128)x->l[i] + y->l[i]; z->l[i] = (uint64_t)c; c >>= 64; }
    return (uint64_t)c;
}
static inline uint64_t fe_sub_raw(fe *z, const fe *x, const fe *y) {
    uint64_t b = 0;
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)x->l[i] - y->l[i] - b;
        z->l[i] = (uint64_t)d; b = (uint64_t)(d >> 64) & 1;
    }
    return b;
}
static inline void fe_add(fe *z, const fe *x, const fe *y, const fctx *F) {
    fe t; fe_add_raw(&t, x, y);       /* p < 2^254 so no carry out */
    if (fe_geq(&t, &F->p)) fe_sub_raw(&t, &t, &F->p);
    *z = t;
}
static inline void fe_sub(fe *z, const fe *x, const fe *y, const fctx *F) {
    fe t; if (fe_sub_raw(&t, x, y)) fe_add_raw(&t, &t, &F->p);
    *z = t;
}
static inline void fe_neg(fe *z, const fe *x, const fctx *F) {
    if (fe_is_zero(x)) { *z = *x; return; }
    fe t; fe_sub_raw(&t, &F->p, x); *z = t;
}
/* CIOS Montgomery product z = x*y/R mod p */
static inline void fe_mul(fe *z, const fe *x, const fe *y, const fctx *F) {
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) {
            c += (u128)x->l[j] * y->l[i] + t[j];
            t[j] = (uint64_t)c; c >>= 64;
        }
        c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
        uint64_t m = t[0] * F->inv;
        c = (u128)m * F->p.l[0] + t[0]; c >>= 64;
        for (int j = 1; j < 4; j++) {
            c += (u128)m * F->p.l[j] + t[j];
            t[j - 1] = (uint64_t)c; c >>= 64;
        }
        c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);
    }
    fe r = {{t[0], t[1], t[2], t[3]}};
    if (t[4] || fe_geq(&r, &F->p)) fe_sub_raw(&r, &r, &F->p);
    *z = r;
}
static inline void fe_sqr(fe *z, const fe *x, const fctx *F) { fe_mul(z, x, x, F); }
static inline void fe_to_mont(fe *z, const fe *x, const fctx *F) { fe_mul(z, x, &F->r2, F); }
static inline void fe_from_mont(fe *z, const fe *x, const fctx *F) {
    fe one = {{1, 0, 0, 0}}; fe_mul(z, x, &one, F);
}
static inline void fe_dbl(fe *z, const fe *x, const fctx *F) { fe_add(z, x, x, F); }
/* z = x^e, e a plain 256-bit integer (4 limbs LE) */
static inline void fe_pow(fe *z, const fe *x, const uint64_t e[4], const fctx *F) {
    fe acc = F->one, b = *x;
    int started = 0;
    for (int i = 255; i >= 0; i--) {
        if (started) fe_sqr(&acc, &acc, F);
        if ((e[i >> 6] >> (i & 63)) & 1) { fe_mul(&acc, &acc, &b, F); started = 1; }
    }
    *z = acc;
}
static inline void fe_inv(fe *z, const fe *x, const fctx *F) { /* Fermat; 0 -> 0 */
    fe e; fe two = {{2, 0, 0, 0}}; fe_sub_raw(&e, &F->p, &two);
    fe_pow(z, x, e.l, F);
}
static inline void fe_set_u64(fe *z, uint64_t v, const fctx *F) {
    fe t = {{v, 0, 0, 0}}; fe_to_mont(z, &t, F);
}
#endif
