/* CPU restatement of the Groth16 prove path on BN254 (the checker and the timed CPU baseline).
 *
 * TEST INFRASTRUCTURE ONLY: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this library; the product (gnark-whir_amd/) never links or calls it.
 *
 * PARITY UNPINNED: the reference runs this path inside un-vendored third-party Go modules
 * (gnark v0.11.0 backend/groth16/bn254/prove.go; gnark-crypto v0.14.1-0.20241217131346-
 * b998989abdbe ecc/bn254/{multiexp.go,fr/fft,marshal.go}; /root/reference/go.mod:6-7) reached
 * from /root/reference/mt.go:496; neither the modules nor a Go toolchain nor any reference
 * test vector exist here.  Each function below restates the published behaviour of the named
 * gnark function and is checked against oracle/pyref.py (big-integer definitions) in
 * tests/test_oracle.py.
 *
 * Same algorithm class as gnark-crypto's CPU path (4x64 Montgomery CIOS, signed-digit bucket
 * MSM, radix-2 DIF/DIT NTT), multi-threaded with OpenMP.  Shares only the public value types
 * of include/mi355x_groth16.h with the product.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <omp.h>
#include "../include/mi355x_groth16.h"
#include "../include/mi355x_groth16_debug.h"   /* MI_DIST_*: the synthetic-input generators are shared with the tests */
#include "ref_field.h"

/* ---------------------------------------------------------------- Fp2 = Fp[u]/(u^2+1) */
typedef struct { fe a0, a1; } fe2;
static inline void fe2_add(fe2 *z, const fe2 *x, const fe2 *y) { fe_add(&z->a0, &x->a0, &y->a0, &FP); fe_add(&z->a1, &x->a1, &y->a1, &FP); }
static inline void fe2_sub(fe2 *z, const fe2 *x, const fe2 *y) { fe_sub(&z->a0, &x->a0, &y->a0, &FP); fe_sub(&z->a1, &x->a1, &y->a1, &FP); }
static inline void fe2_neg(fe2 *z, const fe2 *x) { fe_neg(&z->a0, &x->a0, &FP); fe_neg(&z->a1, &x->a1, &FP); }
static inline void fe2_mul(fe2 *z, const fe2 *x, const fe2 *y) {
    fe t0, t1, t2, t3;
    fe_mul(&t0, &x->a0, &y->a0, &FP); fe_mul(&t1, &x->a1, &y->a1, &FP);
    fe_mul(&t2, &x->a0, &y->a1, &FP); fe_mul(&t3, &x->a1, &y->a0, &FP);
    fe_sub(&z->a0, &t0, &t1, &FP); fe_add(&z->a1, &t2, &t3, &FP);
}
static inline void fe2_sqr(fe2 *z, const fe2 *x) { fe2_mul(z, x, x); }
static inline void fe2_inv(fe2 *z, const fe2 *x) {
    fe n, t; fe_sqr(&n, &x->a0, &FP); fe_sqr(&t, &x->a1, &FP); fe_add(&n, &n, &t, &FP); fe_inv(&n, &n, &FP);
    fe_mul(&z->a0, &x->a0, &n, &FP); fe_mul(&t, &x->a1, &n, &FP); fe_neg(&z->a1, &t, &FP);
}
static inline int fe2_is_zero(const fe2 *x) { return fe_is_zero(&x->a0) && fe_is_zero(&x->a1); }
static inline int fe2_eq(const fe2 *x, const fe2 *y) { return fe_eq(&x->a0, &y->a0) && fe_eq(&x->a1, &y->a1); }

static fe G1_B_MONT;        /* 3 */
static fe2 G2_B_MONT;       /* 3/(9+u) */
static fe2 FE2_ONE;
static int g_init_done = 0;

/* ---------------------------------------------------------------- curve instantiations */
#define EC g1
#define T fe
#define T_ADD(z, x, y) fe_add(z, x, y, &FP)
#define T_SUB(z, x, y) fe_sub(z, x, y, &FP)
#define T_MUL(z, x, y) fe_mul(z, x, y, &FP)
#define T_SQR(z, x) fe_sqr(z, x, &FP)
#define T_NEG(z, x) fe_neg(z, x, &FP)
#define T_INV(z, x) fe_inv(z, x, &FP)
#define T_ISZERO(x) fe_is_zero(x)
#define T_EQ(x, y) fe_eq(x, y)
#define T_ONE (&FP.one)
#define CURVE_B (&G1_B_MONT)
#include "ref_ec_tmpl.h"
#undef EC
#undef T
#undef T_ADD
#undef T_SUB
#undef T_MUL
#undef T_SQR
#undef T_NEG
#undef T_INV
#undef T_ISZERO
#undef T_EQ
#undef T_ONE
#undef CURVE_B

#define EC g2
#define T fe2
#define T_ADD(z, x, y) fe2_add(z, x, y)
#define T_SUB(z, x, y) fe2_sub(z, x, y)
#define T_MUL(z, x, y) fe2_mul(z, x, y)
#define T_SQR(z, x) fe2_sqr(z, x)
#define T_NEG(z, x) fe2_neg(z, x)
#define T_INV(z, x) fe2_inv(z, x)
#define T_ISZERO(x) fe2_is_zero(x)
#define T_EQ(x, y) fe2_eq(x, y)
#define T_ONE (&FE2_ONE)
#define CURVE_B (&G2_B_MONT)
#include "ref_ec_tmpl.h"

static g2_aff G2_GEN_MONT;

static void hex_to_fe_mont(fe *z, const char *dec_unused, const uint64_t l[4]) {
    (void)dec_unused; fe t = {{l[0], l[1], l[2], l[3]}}; fe_to_mont(z, &t, &FP);
}

static void ref_init(void) {
#pragma omp critical(ref_init_lock)
    {
        if (!g_init_done) {
            fe_set_u64(&G1_B_MONT, 3, &FP);
            FE2_ONE.a0 = FP.one; memset(&FE2_ONE.a1, 0, sizeof(fe));
            fe2 nine_u; fe_set_u64(&nine_u.a0, 9, &FP); nine_u.a1 = FP.one;
            fe2 inv; fe2_inv(&inv, &nine_u);
            fe2 three; fe_set_u64(&three.a0, 3, &FP); memset(&three.a1, 0, sizeof(fe));
            fe2_mul(&G2_B_MONT, &three, &inv);
            /* standard BN254 G2 generator (EIP-197 constants), canonical limbs LE */
            static const uint64_t gx0[4] = {0x46debd5cd992f6edULL, 0x674322d4f75edaddULL, 0x426a00665e5c4479ULL, 0x1800deef121f1e76ULL};
            static const uint64_t gx1[4] = {0x97e485b7aef312c2ULL, 0xf1aa493335a9e712ULL, 0x7260bfb731fb5d25ULL, 0x198e9393920d483aULL};
            static const uint64_t gy0[4] = {0x4ce6cc0166fa7daaULL, 0xe3d1e7690c43d37bULL, 0x4aab71808dcb408fULL, 0x12c85ea5db8c6debULL};
            static const uint64_t gy1[4] = {0x55acdadcd122975bULL, 0xbc4b313370b38ef3ULL, 0xec9e99ad690c3395ULL, 0x090689d0585ff075ULL};
            hex_to_fe_mont(&G2_GEN_MONT.x.a0, 0, gx0); hex_to_fe_mont(&G2_GEN_MONT.x.a1, 0, gx1);
            hex_to_fe_mont(&G2_GEN_MONT.y.a0, 0, gy0); hex_to_fe_mont(&G2_GEN_MONT.y.a1, 0, gy1);
            g_init_done = 1;
        }
    }
}

/* ---------------------------------------------------------------- seeded generators (shared definition with the HIP bench generators) */
static inline uint64_t sm64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static inline uint64_t rnd(uint64_t seed, uint64_t idx, uint64_t k) { return sm64(seed ^ sm64(idx * 8 + k)); }
/* 254-bit value, one conditional subtraction of p (2^254 < 2p for both fields) */
static inline void rnd_fe(fe *z, uint64_t seed, uint64_t idx, const fctx *F) {
    for (int k = 0; k < 4; k++) z->l[k] = rnd(seed, idx, k);
    z->l[3] &= 0x3FFFFFFFFFFFFFFFULL;
    if (fe_geq(z, &F->p)) fe_sub_raw(z, z, &F->p);
}

/* canonical scalar of the synthetic workload */
static void gen_scalar_canonical(fe *z, uint64_t seed, uint64_t i, int dist) {
    if (dist == MI_DIST_UNIFORM) { rnd_fe(z, seed, i, &FR); return; }
    /* MI_DIST_WHIR: 45 / 25 / 5 per cent; MI_DIST_MIX(bit, byte, u64): per-mille shares packed into dist (include/mi355x_groth16_debug.h) */
    int mix = (dist & MI_DIST_MIX_FLAG) != 0;
    uint64_t t0 = mix ? (uint64_t)((dist >> 20) & 1023) : 45, t1 = t0 + (mix ? (uint64_t)((dist >> 10) & 1023) : 25), t2 = t1 + (mix ? (uint64_t)(dist & 1023) : 5);
    uint64_t u = rnd(seed, i, 4) % (mix ? 1000 : 100);
    memset(z, 0, sizeof(*z));
    if (u < t0) z->l[0] = rnd(seed, i, 5) & 1;
    else if (u < t1) z->l[0] = rnd(seed, i, 5) & 255;
    else if (u < t2) z->l[0] = rnd(seed, i, 5);
    else rnd_fe(z, seed, i, &FR);
}
void ref_gen_scalars(mi_fr *out, size_t n, uint64_t seed, int dist) { /* Montgomery out */
    ref_init();
#pragma omp parallel for
    for (size_t i = 0; i < n; i++) { fe t; gen_scalar_canonical(&t, seed, i, dist); fe_to_mont((fe *)&out[i], &t, &FR); }
}
void ref_gen_g1(mi_g1_affine *out, size_t n, uint64_t seed) {
    ref_init();
    static const uint64_t e[4] = {0x4f082305b61f3f52ULL, 0x65e05aa45a1c72a3ULL, 0x6e14116da0605617ULL, 0x0c19139cb84c680aULL}; /* (q+1)/4 */
#pragma omp parallel for
    for (size_t i = 0; i < n; i++) {
        fe xc; rnd_fe(&xc, seed, i, &FP);
        fe x, y, rhs, t, one_c = {{1, 0, 0, 0}};
        for (;;) {
            fe_to_mont(&x, &xc, &FP);
            fe_sqr(&rhs, &x, &FP); fe_mul(&rhs, &rhs, &x, &FP); fe_add(&rhs, &rhs, &G1_B_MONT, &FP);
            fe_pow(&y, &rhs, e, &FP);
            fe_sqr(&t, &y, &FP);
            if (fe_eq(&t, &rhs)) break;
            fe_add_raw(&xc, &xc, &one_c);
            if (fe_geq(&xc, &FP.p)) fe_sub_raw(&xc, &xc, &FP.p);
        }
        if (rnd(seed, i, 5) & 1) fe_neg(&y, &y, &FP);
        memcpy(&out[i].x, &x, 32); memcpy(&out[i].y, &y, 32);
    }
}
void ref_gen_g2(mi_g2_affine *out, size_t n, uint64_t seed) { /* k_i * G2, k_i 64-bit odd */
    ref_init();
#pragma omp parallel for
    for (size_t i = 0; i < n; i++) {
        uint64_t k[4] = {rnd(seed, i, 0) | 1, 0, 0, 0};
        g2_jac g, r; g2_jac_from_aff(&g, &G2_GEN_MONT);
        g2_jac_scalar_mul(&r, &g, k);
        g2_aff a; g2_jac_to_aff(&a, &r);
        memcpy(&out[i], &a, sizeof(a));
    }
}

/* ---------------------------------------------------------------- elementwise field / curve ops (parity of the device field layer) */
int32_t ref_field_op(int field, int op, void *zv, const void *xv, const void *yv, size_t n) {
    ref_init();
    const fctx *F = field == 0 ? &FR : &FP;
    fe *z = (fe *)zv; const fe *x = (const fe *)xv; const fe *y = (const fe *)yv;
#pragma omp parallel for
    for (size_t i = 0; i < n; i++) {
        switch (op) {
        case 0: fe_add(&z[i], &x[i], &y[i], F); break;
        case 1: fe_sub(&z[i], &x[i], &y[i], F); break;
        case 2: fe_mul(&z[i], &x[i], &y[i], F); break;
        case 3: fe_inv(&z[i], &x[i], F); break;
        case 4: fe_to_mont(&z[i], &x[i], F); break;
        case 5: fe_from_mont(&z[i], &x[i], F); break;
        }
    }
    return op >= 0 && op <= 5 ? MI_OK : MI_EINVAL;
}
int32_t ref_g1_add(mi_g1_affine *out, const mi_g1_affine *a, const mi_g1_affine *b, size_t n) {
    ref_init();
#pragma omp parallel for
    for (size_t i = 0; i < n; i++) {
        g1_jac p, r; g1_jac_from_aff(&p, (const g1_aff *)&a[i]);
        g1_jac_add_mixed(&r, &p, (const g1_aff *)&b[i], 0);
        g1_jac_to_aff((g1_aff *)&out[i], &r);
    }
    return MI_OK;
}
int32_t ref_g2_add(mi_g2_affine *out, const mi_g2_affine *a, const mi_g2_affine *b, size_t n) {
    ref_init();
#pragma omp parallel for
    for (size_t i = 0; i < n; i++) {
        g2_jac p, r; g2_jac_from_aff(&p, (const g2_aff *)&a[i]);
        g2_jac_add_mixed(&r, &p, (const g2_aff *)&b[i], 0);
        g2_jac_to_aff((g2_aff *)&out[i], &r);
    }
    return MI_OK;
}
int32_t ref_g1_on_curve(const mi_g1_affine *p, size_t n) { ref_init(); for (size_t i = 0; i < n; i++) if (!g1_aff_on_curve((const g1_aff *)&p[i])) return 0; return 1; }
int32_t ref_g2_on_curve(const mi_g2_affine *p, size_t n) { ref_init(); for (size_t i = 0; i < n; i++) if (!g2_aff_on_curve((const g2_aff *)&p[i])) return 0; return 1; }
/* out = k * p, k canonical integer limbs */
int32_t ref_g1_scalar_mul(mi_g1_affine *out, const mi_g1_affine *p, const uint64_t k[4]) {
    ref_init(); g1_jac j, r; g1_jac_from_aff(&j, (const g1_aff *)p); g1_jac_scalar_mul(&r, &j, k); g1_jac_to_aff((g1_aff *)out, &r); return MI_OK;
}
int32_t ref_g2_scalar_mul(mi_g2_affine *out, const mi_g2_affine *p, const uint64_t k[4]) {
    ref_init(); g2_jac j, r; g2_jac_from_aff(&j, (const g2_aff *)p); g2_jac_scalar_mul(&r, &j, k); g2_jac_to_aff((g2_aff *)out, &r); return MI_OK;
}
void ref_g2_generator(mi_g2_affine *out) { ref_init(); memcpy(out, &G2_GEN_MONT, sizeof(*out)); }

/* ---------------------------------------------------------------- fft.Domain (gnark-crypto fr/fft by behaviour) */
static const uint64_t FR_ROOT28[4] = {0x9bd61b6e725b19f0ULL, 0x402d111e41112ed4ULL, 0x00e0a7eb8ef62abcULL, 0x2a3c09f0a58a7e85ULL};

static inline uint32_t bitrev32(uint32_t i, uint32_t logn) {
    uint32_t r = 0; for (uint32_t k = 0; k < logn; k++) { r = (r << 1) | (i & 1); i >>= 1; } return r;
}
static void domain_gen(fe *gen, uint32_t log_n) {
    fe root = {{FR_ROOT28[0], FR_ROOT28[1], FR_ROOT28[2], FR_ROOT28[3]}};
    fe_to_mont(gen, &root, &FR);
    for (uint32_t k = log_n; k < 28; k++) fe_sqr(gen, gen, &FR);
}
/* tw[j] = w^j, j < n/2 */
static fe *build_twiddles(const fe *w, size_t half) {
    fe *tw = (fe *)malloc(sizeof(fe) * (half ? half : 1));
    tw[0] = FR.one;
    for (size_t j = 1; j < half; j++) fe_mul(&tw[j], &tw[j - 1], w, &FR);
    return tw;
}
static void dif_inplace(fe *a, size_t n, const fe *tw) { /* natural in, bit-reversed out */
    for (size_t m = n, step = 1; m >= 2; m >>= 1, step <<= 1) {
        size_t half = m >> 1;
#pragma omp parallel for
        for (size_t k = 0; k < n / 2; k++) {
            size_t blk = k / half, j = k % half, i0 = blk * m + j, i1 = i0 + half;
            fe x = a[i0], y = a[i1], d;
            fe_add(&a[i0], &x, &y, &FR);
            fe_sub(&d, &x, &y, &FR);
            fe_mul(&a[i1], &d, &tw[j * step], &FR);
        }
    }
}
static void dit_inplace(fe *a, size_t n, const fe *tw) { /* bit-reversed in, natural out */
    for (size_t m = 2; m <= n; m <<= 1) {
        size_t half = m >> 1, step = n / m;
#pragma omp parallel for
        for (size_t k = 0; k < n / 2; k++) {
            size_t blk = k / half, j = k % half, i0 = blk * m + j, i1 = i0 + half;
            fe x = a[i0], y;
            fe_mul(&y, &a[i1], &tw[j * step], &FR);
            fe_add(&a[i0], &x, &y, &FR);
            fe_sub(&a[i1], &x, &y, &FR);
        }
    }
}
/* scale slot i by base^(idx(i)) * extra, idx = i or bitrev(i) */
static void scale_pow(fe *a, size_t n, uint32_t log_n, const fe *base, const fe *extra, int bitrev_idx) {
    /* powers via a table of base^(2^k) */
    fe pw[32]; pw[0] = *base; for (int k = 1; k < 32; k++) fe_sqr(&pw[k], &pw[k - 1], &FR);
#pragma omp parallel for
    for (size_t i = 0; i < n; i++) {
        uint32_t e = bitrev_idx ? bitrev32((uint32_t)i, log_n) : (uint32_t)i;
        fe f = *extra;
        for (int k = 0; e; k++, e >>= 1) if (e & 1) fe_mul(&f, &f, &pw[k], &FR);
        fe_mul(&a[i], &a[i], &f, &FR);
    }
}
int32_t ref_ntt(mi_fr *inout, uint32_t log_n, uint32_t flags) {
    ref_init();
    if (log_n > 28 || !inout) return MI_EINVAL;
    fe *a = (fe *)inout; size_t n = (size_t)1 << log_n;
    fe gen, w, g, ginv, ninv, t;
    domain_gen(&gen, log_n);
    fe_set_u64(&g, 5, &FR); fe_inv(&ginv, &g, &FR);
    fe_set_u64(&t, (uint64_t)n, &FR); fe_inv(&ninv, &t, &FR);
    int inverse = flags & MI_NTT_INVERSE, coset = flags & MI_NTT_COSET, dit = flags & MI_NTT_DIT;
    if (inverse) fe_inv(&w, &gen, &FR); else w = gen;
    fe *tw = build_twiddles(&w, n / 2);
    if (!inverse && coset) scale_pow(a, n, log_n, &g, &FR.one, dit ? 1 : 0);
    if (dit) dit_inplace(a, n, tw); else dif_inplace(a, n, tw);
    if (inverse) {
        if (coset) scale_pow(a, n, log_n, &ginv, &ninv, dit ? 0 : 1);
        else {
#pragma omp parallel for
            for (size_t i = 0; i < n; i++) fe_mul(&a[i], &a[i], &ninv, &FR);
        }
    }
    free(tw);
    return MI_OK;
}
/* gnark computeH: h bit-reversed, 2^log_n elements */
int32_t ref_compute_h(uint32_t log_n, const mi_fr *a_in, const mi_fr *b_in, const mi_fr *c_in,
                      size_t n_constraints, mi_fr *h_out) {
    ref_init();
    size_t n = (size_t)1 << log_n;
    if (log_n > 28 || n_constraints > n) return MI_EINVAL;
    fe *a = (fe *)h_out, *b = (fe *)calloc(n, sizeof(fe)), *c = (fe *)calloc(n, sizeof(fe));
    memset(a, 0, n * sizeof(fe));
    memcpy(a, a_in, n_constraints * sizeof(fe)); memcpy(b, b_in, n_constraints * sizeof(fe)); memcpy(c, c_in, n_constraints * sizeof(fe));
    ref_ntt((mi_fr *)a, log_n, MI_NTT_INVERSE); ref_ntt((mi_fr *)b, log_n, MI_NTT_INVERSE); ref_ntt((mi_fr *)c, log_n, MI_NTT_INVERSE);
    ref_ntt((mi_fr *)a, log_n, MI_NTT_DIT | MI_NTT_COSET); ref_ntt((mi_fr *)b, log_n, MI_NTT_DIT | MI_NTT_COSET); ref_ntt((mi_fr *)c, log_n, MI_NTT_DIT | MI_NTT_COSET);
    fe g, den; fe_set_u64(&g, 5, &FR);
    uint64_t e[4] = {(uint64_t)n, 0, 0, 0};
    fe_pow(&den, &g, e, &FR); fe_sub(&den, &den, &FR.one, &FR); fe_inv(&den, &den, &FR);
#pragma omp parallel for
    for (size_t i = 0; i < n; i++) {
        fe t; fe_mul(&t, &a[i], &b[i], &FR); fe_sub(&t, &t, &c[i], &FR); fe_mul(&a[i], &t, &den, &FR);
    }
    ref_ntt((mi_fr *)a, log_n, MI_NTT_INVERSE | MI_NTT_COSET);
    free(b); free(c);
    return MI_OK;
}

/* ---------------------------------------------------------------- MultiExp */
static int pick_c(size_t n) {
    int c = 4; while (c < 16 && ((size_t)1 << (c + 3)) < n) c++;   /* ~ log2(n) - 3, clamp 4..16 */
    return c;
}
static uint64_t (*canon_scalars(const mi_fr *s, size_t n, uint32_t flags))[4] {
    uint64_t(*out)[4] = (uint64_t(*)[4])malloc(n * 32 + 32);
#pragma omp parallel for
    for (size_t i = 0; i < n; i++) {
        fe t; if (flags & MI_MSM_SCALARS_CANONICAL) t = *(const fe *)&s[i]; else fe_from_mont(&t, (const fe *)&s[i], &FR);
        memcpy(out[i], t.l, 32);
    }
    return out;
}
static void norm_g1(mi_g1_jac *out, const g1_jac *j) {
    g1_aff a; g1_jac_to_aff(&a, j);
    if (g1_jac_is_inf(j)) { g1_jac t; g1_jac_set_inf(&t); memcpy(out, &t, sizeof(t)); return; }
    memcpy(&out->x, &a.x, 32); memcpy(&out->y, &a.y, 32); memcpy(&out->z, &FP.one, 32);
}
static void norm_g2(mi_g2_jac *out, const g2_jac *j) {
    g2_aff a; g2_jac_to_aff(&a, j);
    if (g2_jac_is_inf(j)) { g2_jac t; g2_jac_set_inf(&t); memcpy(out, &t, sizeof(t)); return; }
    memcpy(&out->x, &a.x, 64); memcpy(&out->y, &a.y, 64); memcpy(&out->z, &FE2_ONE, 64);
}
int32_t ref_msm_g1(const mi_g1_affine *pts, const mi_fr *scalars, size_t n, uint32_t flags, mi_g1_jac *out) {
    ref_init();
    uint64_t(*sc)[4] = canon_scalars(scalars, n, flags);
    g1_jac r; g1_msm(&r, (const g1_aff *)pts, (const uint64_t(*)[4])sc, n, pick_c(n));
    free(sc); norm_g1(out, &r); return MI_OK;
}
int32_t ref_msm_g2(const mi_g2_affine *pts, const mi_fr *scalars, size_t n, uint32_t flags, mi_g2_jac *out) {
    ref_init();
    uint64_t(*sc)[4] = canon_scalars(scalars, n, flags);
    g2_jac r; g2_msm(&r, (const g2_aff *)pts, (const uint64_t(*)[4])sc, n, pick_c(n));
    free(sc); norm_g2(out, &r); return MI_OK;
}
/* naive sum_i s_i*P_i by double-and-add: the definition, for small n */
int32_t ref_msm_g1_naive(const mi_g1_affine *pts, const mi_fr *scalars, size_t n, uint32_t flags, mi_g1_jac *out) {
    ref_init();
    uint64_t(*sc)[4] = canon_scalars(scalars, n, flags);
    g1_jac acc; g1_jac_set_inf(&acc);
    for (size_t i = 0; i < n; i++) { g1_jac p, t; g1_jac_from_aff(&p, (const g1_aff *)&pts[i]); g1_jac_scalar_mul(&t, &p, sc[i]); g1_jac_add(&acc, &acc, &t); }
    free(sc); norm_g1(out, &acc); return MI_OK;
}
int32_t ref_msm_g2_naive(const mi_g2_affine *pts, const mi_fr *scalars, size_t n, uint32_t flags, mi_g2_jac *out) {
    ref_init();
    uint64_t(*sc)[4] = canon_scalars(scalars, n, flags);
    g2_jac acc; g2_jac_set_inf(&acc);
    for (size_t i = 0; i < n; i++) { g2_jac p, t; g2_jac_from_aff(&p, (const g2_aff *)&pts[i]); g2_jac_scalar_mul(&t, &p, sc[i]); g2_jac_add(&acc, &acc, &t); }
    free(sc); norm_g2(out, &acc); return MI_OK;
}
int32_t ref_g1_sum(const mi_g1_jac *parts, size_t n, mi_g1_jac *out) {
    ref_init(); g1_jac acc; g1_jac_set_inf(&acc);
    for (size_t i = 0; i < n; i++) g1_jac_add(&acc, &acc, (const g1_jac *)&parts[i]);
    norm_g1(out, &acc); return MI_OK;
}

/* ---------------------------------------------------------------- BSB22 Pedersen key (gnark-crypto ecc/bn254/fr/pedersen by behaviour; SURVEY 8f N1;
 * triggered by /root/reference/utilities/utilities.go:189).  Commit / ProveKnowledge are MultiExps over Basis / BasisExpSigma;
 * Fold = sum_i challenge^i * points[i]. */
int32_t ref_pedersen_msm(const mi_g1_affine *bases, const mi_fr *values, size_t n, mi_g1_affine *out) {
    ref_init();
    mi_g1_jac j; ref_msm_g1(bases, values, n, 0, &j);
    g1_jac_to_aff((g1_aff *)out, (const g1_jac *)&j);
    return MI_OK;
}
int32_t ref_pedersen_fold(const mi_g1_affine *points, size_t n, const mi_fr *challenge, mi_g1_affine *out) {
    ref_init();
    fe pw = FR.one; g1_jac acc; g1_jac_set_inf(&acc);
    for (size_t i = 0; i < n; i++) {
        fe k; fe_from_mont(&k, &pw, &FR);
        g1_jac p, t; g1_jac_from_aff(&p, (const g1_aff *)&points[i]); g1_jac_scalar_mul(&t, &p, k.l); g1_jac_add(&acc, &acc, &t);
        fe_mul(&pw, &pw, (const fe *)challenge, &FR);
    }
    g1_jac_to_aff((g1_aff *)out, &acc);
    return MI_OK;
}

/* ---------------------------------------------------------------- BatchScalarMultiplicationG1/G2 (gnark-crypto ecc/bn254 by behaviour; groth16.Setup, mt.go:448) */
int32_t ref_batch_scalar_mul_g1(const mi_g1_affine *base, const mi_fr *scalars, size_t n, mi_g1_affine *out) {
    ref_init();
    g1_jac b; g1_jac_from_aff(&b, (const g1_aff *)base);
#pragma omp parallel for
    for (size_t i = 0; i < n; i++) {
        fe k; fe_from_mont(&k, (const fe *)&scalars[i], &FR);
        g1_jac r; g1_jac_scalar_mul(&r, &b, k.l); g1_jac_to_aff((g1_aff *)&out[i], &r);
    }
    return MI_OK;
}
int32_t ref_batch_scalar_mul_g2(const mi_g2_affine *base, const mi_fr *scalars, size_t n, mi_g2_affine *out) {
    ref_init();
    g2_jac b; g2_jac_from_aff(&b, (const g2_aff *)base);
#pragma omp parallel for
    for (size_t i = 0; i < n; i++) {
        fe k; fe_from_mont(&k, (const fe *)&scalars[i], &FR);
        g2_jac r; g2_jac_scalar_mul(&r, &b, k.l); g2_jac_to_aff((g2_aff *)&out[i], &r);
    }
    return MI_OK;
}

/* ---------------------------------------------------------------- groth16.Prove after the solve (gnark prove.go by behaviour) */
int32_t ref_groth16_prove(const mi_pk_desc *pk, const mi_fr *W, size_t n_wires,
                          const mi_fr *a, const mi_fr *b, const mi_fr *c, size_t n_constraints,
                          const mi_fr *r_m, const mi_fr *s_m, mi_proof_out *out, mi_fr *h_out_opt) {
    ref_init();
    if (!pk || n_wires != pk->nb_wires || pk->log_n > 28) return MI_EINVAL;
    size_t n = (size_t)1 << pk->log_n;
    if (n_constraints > n || pk->n_g1_z < n - 1) return MI_EINVAL;
    /* step 4: h */
    mi_fr *h = h_out_opt ? h_out_opt : (mi_fr *)malloc(n * sizeof(mi_fr));
    ref_compute_h(pk->log_n, a, b, c, n_constraints, h);
    /* step 5: filters */
    mi_fr *wa = (mi_fr *)malloc((n_wires + 1) * sizeof(mi_fr)), *wb = (mi_fr *)malloc((n_wires + 1) * sizeof(mi_fr)),
          *wk = (mi_fr *)malloc((n_wires + 1) * sizeof(mi_fr));
    size_t na = 0, nbb = 0, nk = 0, ci = 0;
    for (size_t j = 0; j < n_wires; j++) {
        if (!pk->infinity_a[j]) wa[na++] = W[j];
        if (!pk->infinity_b[j]) wb[nbb++] = W[j];
        if (j >= pk->nb_public) {
            while (ci < pk->n_committed && pk->committed_wires[ci] < j) ci++;
            if (ci < pk->n_committed && pk->committed_wires[ci] == j) continue;
            wk[nk++] = W[j];
        }
    }
    int32_t rc = MI_OK;
    if (na != pk->n_g1_a || nbb != pk->n_g1_b || nbb != pk->n_g2_b || nk != pk->n_g1_k) rc = MI_EINVAL;
    if (rc == MI_OK) {
        /* step 6: blinding */
        fe rc_, sc_, kr; fe_from_mont(&rc_, (const fe *)r_m, &FR); fe_from_mont(&sc_, (const fe *)s_m, &FR);
        fe_mul(&kr, (const fe *)r_m, (const fe *)s_m, &FR); fe_neg(&kr, &kr, &FR); fe_from_mont(&kr, &kr, &FR);
        g1_jac d1, t, ar, bs1, krs, m;
        g1_jac_from_aff(&d1, (const g1_aff *)&pk->delta1);
        /* Ar = MSM(A, wa) + alpha + r*delta */
        mi_g1_jac mm;
        ref_msm_g1(pk->g1_a, wa, na, 0, &mm); memcpy(&ar, &mm, sizeof(ar));
        g1_jac_add_mixed(&ar, &ar, (const g1_aff *)&pk->alpha1, 0);
        g1_jac_scalar_mul(&t, &d1, rc_.l); g1_jac_add(&ar, &ar, &t);
        /* Bs1 = MSM(B, wb) + beta + s*delta */
        ref_msm_g1(pk->g1_b, wb, nbb, 0, &mm); memcpy(&bs1, &mm, sizeof(bs1));
        g1_jac_add_mixed(&bs1, &bs1, (const g1_aff *)&pk->beta1, 0);
        g1_jac_scalar_mul(&t, &d1, sc_.l); g1_jac_add(&bs1, &bs1, &t);
        /* Krs = MSM(K, wk) + MSM(Z, h[:n-1]) + kr*delta + s*Ar + r*Bs1 */
        ref_msm_g1(pk->g1_k, wk, nk, 0, &mm); memcpy(&krs, &mm, sizeof(krs));
        ref_msm_g1(pk->g1_z, h, n - 1, 0, &mm); memcpy(&m, &mm, sizeof(m));
        g1_jac_add(&krs, &krs, &m);
        g1_jac_scalar_mul(&t, &d1, kr.l); g1_jac_add(&krs, &krs, &t);
        g1_jac_scalar_mul(&t, &ar, sc_.l); g1_jac_add(&krs, &krs, &t);
        g1_jac_scalar_mul(&t, &bs1, rc_.l); g1_jac_add(&krs, &krs, &t);
        /* Bs = MSM_G2(B2, wb) + beta2 + s*delta2 */
        g2_jac bs, d2, t2; mi_g2_jac mm2;
        ref_msm_g2(pk->g2_b, wb, nbb, 0, &mm2); memcpy(&bs, &mm2, sizeof(bs));
        g2_jac_add_mixed(&bs, &bs, (const g2_aff *)&pk->beta2, 0);
        g2_jac_from_aff(&d2, (const g2_aff *)&pk->delta2);
        g2_jac_scalar_mul(&t2, &d2, sc_.l); g2_jac_add(&bs, &bs, &t2);
        g1_jac_to_aff((g1_aff *)&out->ar, &ar);
        g1_jac_to_aff((g1_aff *)&out->krs, &krs);
        g2_jac_to_aff((g2_aff *)&out->bs, &bs);
    }
    free(wa); free(wb); free(wk); if (!h_out_opt) free(h);
    return rc;
}

/* ---------------------------------------------------------------- gnark-crypto point encoding (marshal.go by behaviour) */
static void fe_to_be_bytes(uint8_t out[32], const fe *mont, const fctx *F) {
    fe c; fe_from_mont(&c, mont, F);
    for (int i = 0; i < 4; i++) for (int k = 0; k < 8; k++) out[31 - (8 * i + k)] = (uint8_t)(c.l[i] >> (8 * k));
}
static int fp_lex_largest(const fe *mont) { /* y > (q-1)/2 */
    static const fe half = {{0x9e10460b6c3e7ea3ULL, 0xcbc0b548b438e546ULL, 0xdc2822db40c0ac2eULL, 0x183227397098d014ULL}};
    fe c; fe_from_mont(&c, mont, &FP);
    return fe_geq(&c, &half) && !fe_eq(&c, &half);
}
void ref_g1_compress(const mi_g1_affine *p, uint8_t out[32]) {
    ref_init();
    if (g1_aff_is_inf((const g1_aff *)p)) { memset(out, 0, 32); out[0] = 0x40; return; }
    fe_to_be_bytes(out, (const fe *)&p->x, &FP);
    out[0] |= fp_lex_largest((const fe *)&p->y) ? 0xC0 : 0x80;
}
void ref_g2_compress(const mi_g2_affine *p, uint8_t out[64]) {
    ref_init();
    if (g2_aff_is_inf((const g2_aff *)p)) { memset(out, 0, 64); out[0] = 0x40; return; }
    fe_to_be_bytes(out, (const fe *)&p->x.a1, &FP);
    fe_to_be_bytes(out + 32, (const fe *)&p->x.a0, &FP);
    int largest = fe_is_zero((const fe *)&p->y.a1) ? fp_lex_largest((const fe *)&p->y.a0) : fp_lex_largest((const fe *)&p->y.a1);
    out[0] |= largest ? 0xC0 : 0x80;
}
size_t ref_proof_write(const mi_proof_out *proof, const mi_g1_affine *commitments, uint32_t n_commitments,
                       const mi_g1_affine *pok, uint8_t *out) {
    uint8_t *p = out;
    ref_g1_compress(&proof->ar, p); p += 32;
    ref_g2_compress(&proof->bs, p); p += 64;
    ref_g1_compress(&proof->krs, p); p += 32;
    p[0] = (uint8_t)(n_commitments >> 24); p[1] = (uint8_t)(n_commitments >> 16); p[2] = (uint8_t)(n_commitments >> 8); p[3] = (uint8_t)n_commitments; p += 4;
    for (uint32_t i = 0; i < n_commitments; i++) { ref_g1_compress(&commitments[i], p); p += 32; }
    if (pok) ref_g1_compress(pok, p); else { memset(p, 0, 32); p[0] = 0x40; }
    p += 32;
    return (size_t)(p - out);
}
int32_t ref_num_threads(void) { return omp_get_max_threads(); }
void ref_set_threads(int n) { omp_set_num_threads(n); }
