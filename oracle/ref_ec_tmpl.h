/* CPU restatement, curve layer, "template" instantiated twice by groth16_ref.c:
 *   G1 over Fp  (EC = g1, T = fe)     y^2 = x^3 + 3
 *   G2 over Fp2 (EC = g2, T = fe2)    y^2 = x^3 + 3/(9+u)
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (see ref_field.h).  Restates, by behaviour,
 * gnark-crypto ecc/bn254 g1.go / g2.go / multiexp.go (Jacobian group law, signed-digit
 * bucket method; SURVEY.md section 3.3 "MultiExp internals") as reached from
 * /root/reference/mt.go:496.  Deliberately uses Jacobian buckets (the HIP path uses
 * XYZZ) so that the two implementations do not share formulas.
 *
 * Required macros: EC (name prefix), T (coordinate type), T_ADD/T_SUB/T_MUL/T_SQR/T_NEG/
 * T_INV(z,x[,y]), T_ISZERO(x), T_EQ(x,y), T_ONE (const T*), CURVE_B (const T*).
 */
#define CAT_(a, b) a##_##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(EC, name)
#define AFF CAT(EC, aff)
#define JAC CAT(EC, jac)

typedef struct { T x, y; } AFF;       /* infinity = (0,0) */
typedef struct { T x, y, z; } JAC;    /* infinity = z == 0 */

static inline int FN(aff_is_inf)(const AFF *p) { return T_ISZERO(&p->x) && T_ISZERO(&p->y); }
static inline void FN(jac_set_inf)(JAC *p) { p->x = *T_ONE; p->y = *T_ONE; memset(&p->z, 0, sizeof(T)); }
static inline int FN(jac_is_inf)(const JAC *p) { return T_ISZERO(&p->z); }
static inline void FN(jac_from_aff)(JAC *r, const AFF *p) {
    if (FN(aff_is_inf)(p)) { FN(jac_set_inf)(r); return; }
    r->x = p->x; r->y = p->y; r->z = *T_ONE;
}

static void FN(jac_dbl)(JAC *r, const JAC *p) { /* dbl-2009-l, a = 0 */
    if (FN(jac_is_inf)(p)) { *r = *p; return; }
    T A, B, C, D, E, F, t, X3, Y3, Z3;
    T_SQR(&A, &p->x); T_SQR(&B, &p->y); T_SQR(&C, &B);
    T_ADD(&t, &p->x, &B); T_SQR(&t, &t); T_SUB(&t, &t, &A); T_SUB(&t, &t, &C); T_ADD(&D, &t, &t);
    T_ADD(&E, &A, &A); T_ADD(&E, &E, &A);
    T_SQR(&F, &E);
    T_ADD(&t, &D, &D); T_SUB(&X3, &F, &t);
    T_SUB(&t, &D, &X3); T_MUL(&Y3, &E, &t);
    T_ADD(&C, &C, &C); T_ADD(&C, &C, &C); T_ADD(&C, &C, &C); T_SUB(&Y3, &Y3, &C);
    T_MUL(&Z3, &p->y, &p->z); T_ADD(&Z3, &Z3, &Z3);
    r->x = X3; r->y = Y3; r->z = Z3;
}

static void FN(jac_add)(JAC *r, const JAC *p, const JAC *q) { /* add-2007-bl style, all cases */
    if (FN(jac_is_inf)(p)) { *r = *q; return; }
    if (FN(jac_is_inf)(q)) { *r = *p; return; }
    T Z1Z1, Z2Z2, U1, U2, S1, S2, H, R, HH, HHH, V, t, X3, Y3, Z3;
    T_SQR(&Z1Z1, &p->z); T_SQR(&Z2Z2, &q->z);
    T_MUL(&U1, &p->x, &Z2Z2); T_MUL(&U2, &q->x, &Z1Z1);
    T_MUL(&S1, &p->y, &q->z); T_MUL(&S1, &S1, &Z2Z2);
    T_MUL(&S2, &q->y, &p->z); T_MUL(&S2, &S2, &Z1Z1);
    if (T_EQ(&U1, &U2)) {
        if (T_EQ(&S1, &S2)) { FN(jac_dbl)(r, p); return; }
        FN(jac_set_inf)(r); return;
    }
    T_SUB(&H, &U2, &U1); T_SUB(&R, &S2, &S1);
    T_SQR(&HH, &H); T_MUL(&HHH, &H, &HH); T_MUL(&V, &U1, &HH);
    T_SQR(&X3, &R); T_SUB(&X3, &X3, &HHH); T_ADD(&t, &V, &V); T_SUB(&X3, &X3, &t);
    T_SUB(&t, &V, &X3); T_MUL(&Y3, &R, &t); T_MUL(&t, &S1, &HHH); T_SUB(&Y3, &Y3, &t);
    T_MUL(&Z3, &p->z, &q->z); T_MUL(&Z3, &Z3, &H);
    r->x = X3; r->y = Y3; r->z = Z3;
}

static void FN(jac_add_mixed)(JAC *r, const JAC *p, const AFF *q, int negate_q) {
    if (FN(aff_is_inf)(q)) { *r = *p; return; }
    AFF qq = *q;
    if (negate_q) T_NEG(&qq.y, &qq.y);
    if (FN(jac_is_inf)(p)) { r->x = qq.x; r->y = qq.y; r->z = *T_ONE; return; }
    T Z1Z1, U2, S2, H, R, HH, HHH, V, t, X3, Y3, Z3;
    T_SQR(&Z1Z1, &p->z);
    T_MUL(&U2, &qq.x, &Z1Z1);
    T_MUL(&S2, &qq.y, &p->z); T_MUL(&S2, &S2, &Z1Z1);
    if (T_EQ(&p->x, &U2)) {
        if (T_EQ(&p->y, &S2)) { FN(jac_dbl)(r, p); return; }
        FN(jac_set_inf)(r); return;
    }
    T_SUB(&H, &U2, &p->x); T_SUB(&R, &S2, &p->y);
    T_SQR(&HH, &H); T_MUL(&HHH, &H, &HH); T_MUL(&V, &p->x, &HH);
    T_SQR(&X3, &R); T_SUB(&X3, &X3, &HHH); T_ADD(&t, &V, &V); T_SUB(&X3, &X3, &t);
    T_SUB(&t, &V, &X3); T_MUL(&Y3, &R, &t); T_MUL(&t, &p->y, &HHH); T_SUB(&Y3, &Y3, &t);
    T_MUL(&Z3, &p->z, &H);
    r->x = X3; r->y = Y3; r->z = Z3;
}

static void FN(jac_to_aff)(AFF *r, const JAC *p) {
    if (FN(jac_is_inf)(p)) { memset(r, 0, sizeof(*r)); return; }
    T zi, zi2, zi3;
    T_INV(&zi, &p->z); T_SQR(&zi2, &zi); T_MUL(&zi3, &zi2, &zi);
    T_MUL(&r->x, &p->x, &zi2); T_MUL(&r->y, &p->y, &zi3);
}

static int FN(aff_on_curve)(const AFF *p) {
    if (FN(aff_is_inf)(p)) return 1;
    T l, rr;
    T_SQR(&l, &p->y);
    T_SQR(&rr, &p->x); T_MUL(&rr, &rr, &p->x); T_ADD(&rr, &rr, CURVE_B);
    return T_EQ(&l, &rr);
}

/* k*P, k a canonical (non-Montgomery) 256-bit integer */
static void FN(jac_scalar_mul)(JAC *r, const JAC *p, const uint64_t k[4]) {
    JAC acc; FN(jac_set_inf)(&acc);
    for (int i = 255; i >= 0; i--) {
        FN(jac_dbl)(&acc, &acc);
        if ((k[i >> 6] >> (i & 63)) & 1) FN(jac_add)(&acc, &acc, p);
    }
    *r = acc;
}

/* Signed-digit bucket method.  scalars: canonical integers < r, 4 limbs each.
 * OpenMP parallel over windows.  c in [2,16]. */
static void FN(msm)(JAC *out, const AFF *pts, const uint64_t (*sc)[4], size_t n, int c) {
    int nwin = (255 + c - 1) / c;
    size_t nb = (size_t)1 << (c - 1);
    JAC *wsum = (JAC *)malloc(sizeof(JAC) * nwin);
#pragma omp parallel for schedule(dynamic, 1)
    for (int w = 0; w < nwin; w++) {
        JAC *bk = (JAC *)malloc(sizeof(JAC) * nb);
        for (size_t b = 0; b < nb; b++) FN(jac_set_inf)(&bk[b]);
        for (size_t i = 0; i < n; i++) {
            /* recompute the signed digit of window w: digit = raw + carry_in, where carry_in
             * is 1 iff the lower windows produced a carry; evaluate carries from window 0. */
            int carry = 0; int64_t d = 0;
            for (int v = 0; v <= w; v++) {
                int bit = v * c; int limb = bit >> 6, sh = bit & 63;
                uint64_t raw = 0;
                if (limb < 4) {
                    raw = sc[i][limb] >> sh;
                    if (sh + c > 64 && limb + 1 < 4) raw |= sc[i][limb + 1] << (64 - sh);
                }
                raw &= ((uint64_t)1 << c) - 1;
                d = (int64_t)raw + carry;
                if (d > (int64_t)nb) { d -= (int64_t)1 << c; carry = 1; } else carry = 0;
            }
            if (d == 0) continue;
            if (d > 0) FN(jac_add_mixed)(&bk[d - 1], &bk[d - 1], &pts[i], 0);
            else FN(jac_add_mixed)(&bk[-d - 1], &bk[-d - 1], &pts[i], 1);
        }
        JAC run, acc; FN(jac_set_inf)(&run); FN(jac_set_inf)(&acc);
        for (size_t b = nb; b-- > 0;) {
            FN(jac_add)(&run, &run, &bk[b]);
            FN(jac_add)(&acc, &acc, &run);
        }
        wsum[w] = acc;
        free(bk);
    }
    JAC tot; FN(jac_set_inf)(&tot);
    for (int w = nwin - 1; w >= 0; w--) {
        for (int k = 0; k < c; k++) FN(jac_dbl)(&tot, &tot);
        FN(jac_add)(&tot, &tot, &wsum[w]);
    }
    free(wsum);
    *out = tot;
}

#undef AFF
#undef JAC
#undef FN
#undef CAT
#undef CAT_
