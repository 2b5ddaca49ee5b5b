// Host build of the device arithmetic headers (test infrastructure): lets the CPU test-suite run
// the exact field / curve code of gnark-whir_amd/csrc on the host and compare it with the oracle.
#include <cstddef>
#include <cstring>
#include "../../gnark-whir_amd/csrc/curve.cuh"

template <class P>
static void field_op(int op, Fe<P> *z, const Fe<P> *x, const Fe<P> *y, size_t n) {
    for (size_t i = 0; i < n; i++) {
        switch (op) {
        case 0: z[i] = x[i] + y[i]; break;
        case 1: z[i] = x[i] - y[i]; break;
        case 2: z[i] = x[i] * y[i]; break;
        case 3: z[i] = fe_inv(x[i]); break;
        case 4: z[i] = fe_to_mont(x[i]); break;
        case 5: z[i] = fe_from_mont(x[i]); break;
        case 6: z[i] = fe_neg(x[i]); break;
        case 7: z[i] = fe_mul2_add(x[i], y[i], y[i], x[i]); break;            // 2xy/R
        case 8: z[i] = fe_mul_sub(x[i], y[i], y[i], y[i]); break;             // (xy - y^2)/R
        case 9: z[i] = fe_sqr(x[i]); break;
        }
    }
}
template <class F>
static void ec_add(Affine<F> *out, const Affine<F> *a, const Affine<F> *b, size_t n, int mode) {
    for (size_t i = 0; i < n; i++) {
        XYZZ<F> acc = XYZZ<F>::from_affine(a[i]);
        if (mode == 0) xyzz_madd(acc, b[i], false);
        else if (mode == 1) {  // full add with non-trivial zz on both sides: scale by doubling tricks
            XYZZ<F> q = XYZZ<F>::from_affine(b[i]);
            // re-randomise representation: (X l^2, Y l^3, ZZ l^2, ZZZ l^3) with l = x-coordinate+1
            if (!acc.is_inf()) { F l = a[i].x + F::one(); F l2 = fe_sqr(l), l3 = l2 * l; acc = XYZZ<F>{acc.x * l2, acc.y * l3, acc.zz * l2, acc.zzz * l3}; }
            if (!q.is_inf()) { F l = b[i].y + F::one(); F l2 = fe_sqr(l), l3 = l2 * l; q = XYZZ<F>{q.x * l2, q.y * l3, q.zz * l2, q.zzz * l3}; }
            xyzz_add(acc, q);
        } else {  // a - b via negated madd
            xyzz_madd(acc, b[i], true);
        }
        out[i] = xyzz_to_affine(acc);
    }
}
extern "C" {
int emu_field_op(int field, int op, void *z, const void *x, const void *y, size_t n) {
    if (field == 0) field_op<FrParams>(op, (Fr *)z, (const Fr *)x, (const Fr *)y, n);
    else field_op<FpParams>(op, (Fp *)z, (const Fp *)x, (const Fp *)y, n);
    return 0;
}
int emu_g1_add(void *out, const void *a, const void *b, size_t n, int mode) { ec_add<Fp>((G1Aff *)out, (const G1Aff *)a, (const G1Aff *)b, n, mode); return 0; }
int emu_g2_add(void *out, const void *a, const void *b, size_t n, int mode) { ec_add<Fp2>((G2Aff *)out, (const G2Aff *)a, (const G2Aff *)b, n, mode); return 0; }
int emu_g1_mul_u32(void *out, const void *a, uint32_t k) { G1X r = xyzz_mul_u32(G1X::from_affine(*(const G1Aff *)a), k); *(G1Aff *)out = xyzz_to_affine(r); return 0; }
void emu_g2_b(void *out) { Fp2 b = curve_b((const Fp2 *)0); memcpy(out, &b, sizeof(b)); }
}

// ---------------------------------------------------------------- 9 x 29-bit level-1 accumulation (curve29.cuh on the host; traps on any
// violated limb / carry assumption because this file is built with -DMI_CHECK_NOWRAP)
#include <vector>
#include "../../gnark-whir_amd/csrc/curve29.cuh"
extern "C" int emu_g1_madd29_chain(void *out_aff, const void *pts_std, const unsigned char *neg, size_t n) {
    const G1Aff *p = (const G1Aff *)pts_std;
    G1X29 acc = g1x29_inf();
    for (size_t i = 0; i < n; i++) {
        G1Aff rp{fe_to_rprime_packed(p[i].x), fe_to_rprime_packed(p[i].y)};
        u32 w[16];
        memcpy(w, &rp, 64);
        g1x29_madd(acc, w, neg[i] != 0);
    }
    *(G1Aff *)out_aff = xyzz_to_affine(g1x29_to_std(acc));
    return 0;
}
// items of `item_len` mixed additions each, their sums packed in the R' form and added up with g1x29_add (levels >= 2 of the device's
// item machinery): level-1 kernel -> store_rp -> load_rp -> full additions -> standard form
extern "C" int emu_g1_rp_levels(void *out_aff, const void *pts_std, const unsigned char *neg, size_t n, size_t item_len) {
    const G1Aff *p = (const G1Aff *)pts_std;
    std::vector<G1X> partial;
    for (size_t b = 0; b < n; b += item_len) {
        G1X29 acc = g1x29_inf();
        for (size_t i = b; i < n && i < b + item_len; i++) {
            G1Aff rp{fe_to_rprime_packed(p[i].x), fe_to_rprime_packed(p[i].y)};
            u32 w[16];
            memcpy(w, &rp, 64);
            g1x29_madd(acc, w, neg[i] != 0);
        }
        G1X slot;
        g1x29_store_rp(acc, reinterpret_cast<u32 *>(&slot));
        partial.push_back(slot);
    }
    while (partial.size() > 1) {   // levels of up to three partial sums per item
        std::vector<G1X> next;
        for (size_t b = 0; b < partial.size(); b += 3) {
            G1X29 acc = g1x29_load_rp(reinterpret_cast<const u32 *>(&partial[b]));
            for (size_t k = b + 1; k < partial.size() && k < b + 3; k++) g1x29_add(acc, g1x29_load_rp(reinterpret_cast<const u32 *>(&partial[k])));
            G1X slot;
            g1x29_store_rp(acc, reinterpret_cast<u32 *>(&slot));
            next.push_back(slot);
        }
        partial.swap(next);
    }
    G1X29 fin = partial.empty() ? g1x29_inf() : g1x29_load_rp(reinterpret_cast<const u32 *>(&partial[0]));
    *(G1Aff *)out_aff = xyzz_to_affine(g1x29_to_std(fin));
    return 0;
}
#include "../../gnark-whir_amd/csrc/curve29_g2.cuh"
struct RegAccG2_29 { F2_29 c[4]; F2_29 ld(int k) const { return c[k]; } void st(int k, const F2_29 &v) { c[k] = v; } };
extern "C" int emu_g2_madd29_chain(void *out_aff, const void *pts_std, const unsigned char *neg, size_t n) {
    const G2Aff *p = (const G2Aff *)pts_std;
    RegAccG2_29 A{};
    bool inf = true;
    for (size_t i = 0; i < n; i++) {
        G2Aff rp{Fp2{fe_to_rprime_packed(p[i].x.a0), fe_to_rprime_packed(p[i].x.a1)}, Fp2{fe_to_rprime_packed(p[i].y.a0), fe_to_rprime_packed(p[i].y.a1)}};
        u32 w[32];
        memcpy(w, &rp, 128);
        g2x29_madd(A, inf, w, neg[i] != 0);
    }
    G2X s = inf ? G2X::inf() : G2X{f2_29_to_std(A.ld(0)), f2_29_to_std(A.ld(1)), f2_29_to_std(A.ld(2)), f2_29_to_std(A.ld(3))};
    *(G2Aff *)out_aff = xyzz_to_affine(s);
    return 0;
}
// the G2 twin of emu_g1_rp_levels: items of mixed additions, their sums packed (f2_29_pack) and added up with g2x29_add -- what
// k_msm_accum_affine_g2_29 + k_msm_accum_xyzz_g2_29 do on the device
extern "C" int emu_g2_rp_levels(void *out_aff, const void *pts_std, const unsigned char *neg, size_t n, size_t item_len) {
    const G2Aff *p = (const G2Aff *)pts_std;
    struct Slot { u32 w[64]; };
    auto store = [](const RegAccG2_29 &A, bool inf) {
        Slot s;
        memset(&s, 0, sizeof(s));
        if (!inf) for (int c = 0; c < 4; c++) f2_29_pack(A.ld(c), s.w + 16 * c);
        return s;
    };
    auto is_inf = [](const Slot &s) { u32 any = 0; for (int i = 32; i < 48; i++) any |= s.w[i]; return any == 0; };
    std::vector<Slot> partial;
    for (size_t b = 0; b < n; b += item_len) {
        RegAccG2_29 A{};
        bool inf = true;
        for (size_t i = b; i < n && i < b + item_len; i++) {
            G2Aff rp{Fp2{fe_to_rprime_packed(p[i].x.a0), fe_to_rprime_packed(p[i].x.a1)}, Fp2{fe_to_rprime_packed(p[i].y.a0), fe_to_rprime_packed(p[i].y.a1)}};
            u32 w[32];
            memcpy(w, &rp, 128);
            g2x29_madd(A, inf, w, neg[i] != 0);
        }
        partial.push_back(store(A, inf));
    }
    while (partial.size() > 1) {   // levels of up to three partial sums per item
        std::vector<Slot> next;
        for (size_t b = 0; b < partial.size(); b += 3) {
            RegAccG2_29 A{};
            bool inf = true;
            for (size_t k = b; k < partial.size() && k < b + 3; k++) {
                const u32 *bw = partial[k].w;
                g2x29_add(A, inf, [bw](int comp) { return f2_29_unpack(bw + 16 * comp); }, is_inf(partial[k]));
            }
            next.push_back(store(A, inf));
        }
        partial.swap(next);
    }
    G2X s = G2X::inf();
    if (!partial.empty() && !is_inf(partial[0])) {
        const u32 *bw = partial[0].w;
        s = G2X{f2_29_to_std(f2_29_unpack(bw)), f2_29_to_std(f2_29_unpack(bw + 16)), f2_29_to_std(f2_29_unpack(bw + 32)), f2_29_to_std(f2_29_unpack(bw + 48))};
    }
    *(G2Aff *)out_aff = xyzz_to_affine(s);
    return 0;
}
// raw primitives of field29.cuh on limb vectors chosen by the test (9 x u32 each); field 1 = Fp, 0 = Fr
extern "C" int emu_f29_prim(int field, int op, u32 *out, const u32 *a, const u32 *b, const u32 *c, const u32 *d) {
    F29 A, B, C, D, R;
    memcpy(A.l, a, 36); memcpy(B.l, b, 36); memcpy(C.l, c, 36); memcpy(D.l, d, 36);
    if (field == 1) {
        switch (op) {
        case 0: R = f29_mul<FpParams>(A, B); break;
        case 1: R = f29_mul2<FpParams>(A, B, C, D); break;
        case 2: R = f29_sub<FpParams>(A, B, P29<FpParams>::c8); break;
        case 3: R = f29_sub<FpParams>(A, B, P29<FpParams>::c4); break;
        case 4: R = f29_sub<FpParams>(A, B, P29<FpParams>::c2); break;
        case 5: R = f29_wnorm(A); break;
        case 6: R = f29_condsub(A, P29<FpParams>::p4); break;
        case 7: R = f29_condsub(A, P29<FpParams>::p2); break;
        case 8: R = f29_sqr<FpParams>(A); break;
        default: return -1;
        }
    } else {
        switch (op) {
        case 0: R = f29_mul<FrParams>(A, B); break;
        case 1: R = f29_mul2<FrParams>(A, B, C, D); break;
        default: return -1;
        }
    }
    memcpy(out, R.l, 36);
    return 0;
}
extern "C" int emu_f29_roundtrip(void *out_std, const void *in_std, size_t n, int field) {   // std -> R' limbs -> std, and a product in both
    for (size_t i = 0; i < n; i++) {
        if (field == 1) {
            const Fp x = ((const Fp *)in_std)[2 * i], y = ((const Fp *)in_std)[2 * i + 1];
            F29 a = f29_from_std<FpParams>(x), b = f29_from_std<FpParams>(y);
            ((Fp *)out_std)[2 * i] = f29_to_std<FpParams>(a);
            ((Fp *)out_std)[2 * i + 1] = f29_to_std<FpParams>(f29_mul<FpParams>(a, b));
        } else {
            const Fr x = ((const Fr *)in_std)[2 * i], y = ((const Fr *)in_std)[2 * i + 1];
            F29 a = f29_from_std<FrParams>(x), b = f29_from_std<FrParams>(y);
            ((Fr *)out_std)[2 * i] = f29_to_std<FrParams>(a);
            ((Fr *)out_std)[2 * i + 1] = f29_to_std<FrParams>(f29_mul<FrParams>(a, b));
        }
    }
    return 0;
}

// ---------------------------------------------------------------- NTT pass emulation (ntt_tile.cuh on the host)
#include <vector>
#include "../../gnark-whir_amd/csrc/ntt_tile.cuh"
static Fr fr_pow_u64(Fr b, u64 e) { Fr acc = Fr::one(); while (e) { if (e & 1) acc = acc * b; b = fe_sqr(b); e >>= 1; } return acc; }
static Fr fr_from_u64x4(u64 a, u64 b, u64 c, u64 d) {
    Fr t; t.l[0] = (u32)a; t.l[1] = (u32)(a >> 32); t.l[2] = (u32)b; t.l[3] = (u32)(b >> 32);
    t.l[4] = (u32)c; t.l[5] = (u32)(c >> 32); t.l[6] = (u32)d; t.l[7] = (u32)(d >> 32);
    return fe_to_mont(t);
}
// load_mul / store_sub: computeH's fused pointwise steps (NttPass), or null
extern "C" int emu_ntt_fused(void *data_v, uint32_t log_n, uint32_t flags, uint32_t log_e, uint32_t max_contig, uint32_t max_strided,
                             uint32_t nthr, uint32_t n_valid, const void *load_mul, const void *store_sub) {
    Fr *data = (Fr *)data_v;
    const bool inverse = flags & 1, coset = flags & 2, dit = flags & 4;
    Fr root = fr_from_u64x4(0x9bd61b6e725b19f0ull, 0x402d111e41112ed4ull, 0x00e0a7eb8ef62abcull, 0x2a3c09f0a58a7e85ull);
    Fr w2048 = root; for (int k = 12; k < 28; k++) w2048 = fe_sqr(w2048);   // w_4096 (name kept)
    Fr w = root; for (u32 k = log_n; k < 28; k++) w = fe_sqr(w);
    if (inverse) { w = fe_inv(w); w2048 = fe_inv(w2048); }
    Fr g = fe_from_u32<FrParams>(5); if (inverse) g = fe_inv(g);
    Fr nn = Fr::zero(); nn.l[0] = 1u << log_n; Fr ninv = fe_inv(fe_to_mont(nn));
    u32 h = (log_n + 1) / 2, nlo = 1u << h, nhi = 1u << (log_n - h);
    std::vector<Fr> small(2048), twlo(nlo), twhi(nhi), sclo(nlo), schi(nhi);
    for (u32 j = 0; j < 2048; j++) small[j] = fr_pow_u64(w2048, j);
    for (u32 j = 0; j < nlo; j++) { twlo[j] = fr_pow_u64(w, j); sclo[j] = fr_pow_u64(g, j); }
    for (u32 j = 0; j < nhi; j++) { twhi[j] = fr_pow_u64(w, (u64)j << h); schi[j] = fr_pow_u64(g, (u64)j << h); if (inverse) schi[j] = schi[j] * ninv; }
    NttTables t{small.data(), twlo.data(), twhi.data(), sclo.data(), schi.data(), nullptr, h};
    std::vector<Fr> tw64k;
    if (log_n % 2 == 0) {   // exercise the direct-table path on half of the sizes
        Fr w64k = root; for (int k = 16; k < 28; k++) w64k = fe_sqr(w64k);
        if (inverse) w64k = fe_inv(w64k);
        tw64k.resize(65536); tw64k[0] = Fr::one();
        for (u32 j = 1; j < 65536; j++) tw64k[j] = tw64k[j - 1] * w64k;
        t.tw_64k = tw64k.data();
    }
    std::vector<Fr> nv(1, ninv);
    u32 load_scale = 0, store_scale = 0;
    if (coset && !inverse) load_scale = dit ? 1 : 2;
    else if (coset && inverse) store_scale = dit ? 4 : 3;
    else if (inverse) { t.sc_lo = nv.data(); t.sc_hi = nv.data(); store_scale = 5; }
    NttPlan pl = ntt_make_plan(log_n, max_contig, max_strided);
    u32 log_s[8];
    for (u32 i = 0, acc = log_n; i < pl.n_pass; i++) { acc -= pl.log_r[i]; log_s[i] = acc; }
    for (u32 step = 0; step < pl.n_pass; step++) {
        u32 i = dit ? pl.n_pass - 1 - step : step;
        NttPass p{};
        p.log_n = log_n; p.log_r = pl.log_r[i]; p.log_s = log_s[i];
        u32 room = log_e > p.log_r ? log_e - p.log_r : 0;
        u32 avail = p.log_s == 0 ? log_n - p.log_r : p.log_s;
        p.log_c = room < avail ? room : avail;
        p.dit = dit; p.twiddle = p.log_s != 0; p.scale = 0;
        if (step == 0 && load_scale) p.scale = load_scale;
        if (step == pl.n_pass - 1 && store_scale) { if (p.scale) return -1; p.scale = store_scale; }
        p.n_valid = step == 0 ? n_valid : (1u << log_n);
        if (step == 0) p.load_mul = (const Fr *)load_mul;
        if (step == pl.n_pass - 1) { p.store_sub = (const Fr *)store_sub; p.canon = 1; }   // (the elements are lazily reduced until the last store: ntt_tile.cuh)
        u32 tiles = 1u << (log_n - p.log_r - p.log_c);
        std::vector<U4> lds((size_t)2 << (p.log_r + p.log_c));
        for (u32 tile = 0; tile < tiles; tile++) {
            for (u32 tid = 0; tid < nthr; tid++) ntt_tile_load(p, t, data, tile, tid, nthr, lds.data());
            for (u32 s = 0; s < p.log_r; s++)
                for (u32 tid = 0; tid < nthr; tid++) ntt_tile_stage(p, t, s, tid, nthr, lds.data());
            for (u32 tid = 0; tid < nthr; tid++) ntt_tile_store(p, t, data, tile, tid, nthr, lds.data());
        }
    }
    return (int)pl.n_pass;
}
extern "C" int emu_ntt(void *data_v, uint32_t log_n, uint32_t flags, uint32_t log_e, uint32_t max_contig, uint32_t max_strided,
                       uint32_t nthr, uint32_t n_valid) {
    return emu_ntt_fused(data_v, log_n, flags, log_e, max_contig, max_strided, nthr, n_valid, nullptr, nullptr);
}

// ---------------------------------------------------------------- MSM pipeline emulation (msm_core.cuh on the host)
#include "../../gnark-whir_amd/csrc/msm_core.cuh"
static void excl_scan(const std::vector<u32> &in, std::vector<u32> &out) {
    out.resize(in.size() + 1);
    u32 acc = 0;
    for (size_t i = 0; i < in.size(); i++) { out[i] = acc; acc += in[i]; }
    out[in.size()] = acc;
}
// runs the item/level machinery until every key is final; returns number of levels
template <class F>
static int run_levels(u32 nkeys, std::vector<u32> start, std::vector<u32> cnt, std::vector<u32> items, u32 L,
                      const Affine<F> *pts, const u32 *sorted, std::vector<XYZZ<F>> partial, std::vector<XYZZ<F>> &bucket, u32 nthr_unused) {
    int level = 0;
    for (;;) {
        std::vector<u32> item_start;
        excl_scan(items, item_start);
        u32 total = item_start[nkeys];
        std::vector<XYZZ<F>> out(total ? total : 1);
        bool more = false;
        for (u32 k = 0; k < nkeys; k++) more |= items[k] > 1;
        for (u32 it = 0; it < total + 3; it++) {  // over-launch like the GPU grid does
            if (level == 0) msm_accum_affine_body<F>(pts, sorted, start.data(), cnt.data(), items.data(), item_start.data(), nkeys, L, bucket.data(), out.data(), it);
            else msm_accum_xyzz_body<F>(partial.data(), start.data(), cnt.data(), items.data(), item_start.data(), nkeys, L, bucket.data(), out.data(), it);
        }
        level++;
        if (!more) break;
        std::vector<u32> s2(nkeys), c2(nkeys), i2(nkeys);
        for (u32 k = 0; k < nkeys; k++) msm_prep_next(items.data(), item_start.data(), L, s2.data(), c2.data(), i2.data(), k);
        start = s2; cnt = c2; items = i2; partial = out;
    }
    return level;
}
template <class F>
static int emu_msm_t(void *out_v, const void *pts_v, const void *sc_v, u32 n, int mont, u32 c, u32 G, u32 L, u32 seg, u32 nthr) {
    const Affine<F> *pts = (const Affine<F> *)pts_v;
    MsmShape s = msm_shape(n, c, G);
    std::vector<int16_t> digits((size_t)s.nwin * n + 1);
    for (u32 i = 0; i < n; i++) msm_digits_body(s, (const Fr *)sc_v, mont != 0, digits.data(), i);
    std::vector<u32> H((size_t)s.nkeys * G), lds(s.nbuckets);
    for (u32 w = 0; w < s.nwin; w++)
        for (u32 g = 0; g < G; g++) {
            for (u32 t = 0; t < nthr; t++) msm_hist_zero(s, lds.data(), t, nthr);
            for (u32 t = 0; t < nthr; t++) msm_hist_count(s, digits.data(), g, w, lds.data(), t, nthr);
            for (u32 t = 0; t < nthr; t++) msm_hist_write(s, H.data(), g, w, lds.data(), t, nthr);
        }
    std::vector<u32> total(s.nkeys), S;
    for (u32 k = 0; k < s.nkeys; k++) msm_colsum_body(s, H.data(), total.data(), k);
    excl_scan(total, S);
    std::vector<u32> sorted(S.back() + 1);
    for (u32 w = 0; w < s.nwin; w++)
        for (u32 g = 0; g < G; g++) {
            for (u32 t = 0; t < nthr; t++) msm_scatter_init(s, S.data(), H.data(), g, w, lds.data(), t, nthr);
            for (u32 t = 0; t < nthr; t++) msm_scatter_move(s, digits.data(), g, w, lds.data(), sorted.data(), t, nthr);
        }
    std::vector<u32> start(s.nkeys), cnt(s.nkeys), items(s.nkeys);
    for (u32 k = 0; k < s.nkeys; k++) msm_prep_level1(s, S.data(), L, start.data(), cnt.data(), items.data(), k);
    std::vector<XYZZ<F>> bucket(s.nkeys);
    memset(bucket.data(), 0, sizeof(XYZZ<F>) * s.nkeys);  // zz == 0 : infinity
    int levels = run_levels<F>(s.nkeys, start, cnt, items, L, pts, sorted.data(), std::vector<XYZZ<F>>(), bucket, nthr);
    // bucket reduce + reduce by window
    u32 tb = (s.nbuckets + seg - 1) / seg;
    std::vector<XYZZ<F>> P((size_t)s.nwin * tb);
    for (u32 w = 0; w < s.nwin; w++)
        for (u32 t = 0; t < tb; t++) msm_bucket_reduce_body<F>(bucket.data(), s.nbuckets, seg, P.data(), w, t);
    std::vector<u32> ws(s.nwin), wc(s.nwin), wi(s.nwin);
    for (u32 w = 0; w < s.nwin; w++) { ws[w] = w * tb; wc[w] = tb; wi[w] = (tb + L - 1) / L; }
    std::vector<XYZZ<F>> wsum(s.nwin);
    memset(wsum.data(), 0, sizeof(XYZZ<F>) * s.nwin);
    // level machinery over partials: level index starts at 1 (xyzz source)
    {
        std::vector<u32> start2 = ws, cnt2 = wc, items2 = wi;
        std::vector<XYZZ<F>> partial = P;
        for (;;) {
            std::vector<u32> item_start;
            excl_scan(items2, item_start);
            u32 total = item_start[s.nwin];
            std::vector<XYZZ<F>> out(total ? total : 1);
            bool more = false;
            for (u32 k = 0; k < s.nwin; k++) more |= items2[k] > 1;
            for (u32 it = 0; it < total + 2; it++)
                msm_accum_xyzz_body<F>(partial.data(), start2.data(), cnt2.data(), items2.data(), item_start.data(), s.nwin, L, wsum.data(), out.data(), it);
            if (!more) break;
            std::vector<u32> s3(s.nwin), c3(s.nwin), i3(s.nwin);
            for (u32 k = 0; k < s.nwin; k++) msm_prep_next(items2.data(), item_start.data(), L, s3.data(), c3.data(), i3.data(), k);
            start2 = s3; cnt2 = c3; items2 = i3; partial = out;
        }
    }
    XYZZ<F> tot = msm_combine_windows<F>(wsum.data(), s.nwin, s.c);
    *(Affine<F> *)out_v = xyzz_to_affine(tot);
    return levels;
}
extern "C" int emu_msm_g1(void *out, const void *pts, const void *sc, uint32_t n, int mont, uint32_t c, uint32_t G, uint32_t L, uint32_t seg, uint32_t nthr) {
    return emu_msm_t<Fp>(out, pts, sc, n, mont, c, G, L, seg, nthr);
}
extern "C" int emu_msm_g2(void *out, const void *pts, const void *sc, uint32_t n, int mont, uint32_t c, uint32_t G, uint32_t L, uint32_t seg, uint32_t nthr) {
    return emu_msm_t<Fp2>(out, pts, sc, n, mont, c, G, L, seg, nthr);
}

// ---------------------------------------------------------------- fixed-base MSM (msm2_core.cuh on the host)
#include "../../gnark-whir_amd/csrc/msm2_core.cuh"
extern "C" int emu_msm2_g1(void *out_v, const void *pts_v, const void *sc_v, uint32_t n, int mont, uint32_t c, uint32_t G, uint32_t chunk, uint32_t L,
                           uint32_t seg, uint32_t nthr, uint32_t gbits, uint32_t wkeys) {
    typedef Fp F;
    Msm2Shape s = msm2_shape(n, c, G, chunk, gbits, wkeys);
    // pk_load: window copies (fixed-base).  wkeys: the generic MSM borrows the sort, points stay as they are
    std::vector<Affine<F>> pre((size_t)(wkeys ? 1 : s.nwin) * n);
    if (wkeys) memcpy(pre.data(), pts_v, sizeof(Affine<F>) * n);
    else for (u32 i = 0; i < n; i++) msm2_precompute_body<F>((const Affine<F> *)pts_v, pre.data(), n, c, s.nwin, i);
    // pass 1
    std::vector<u32> C1((size_t)s.ngroups * G), lds(s.gsize > s.ngroups ? s.gsize : s.ngroups);
    std::vector<u32> two(2 * (size_t)s.ngroups);
    for (u32 g = 0; g < G; g += 2) {   // two slices per counting workgroup, like k_msm2_count's `per` (the prefetching walk)
        const u32 cnt = g + 2 <= G ? 2 : 1;
        for (auto &x : two) x = 0;
#define EMU_COUNT(C) msm2_count_slices<C>(s, (const Fr *)sc_v, mont != 0, g, cnt, two.data(), t, nthr)
        for (u32 t = 0; t < nthr; t++) { MSM2_FOR_C(s.c, EMU_COUNT) }   // the compile-time-width digits where the kernels use them
#undef EMU_COUNT
        for (u32 j = 0; j < cnt; j++) for (u32 h = 0; h < s.ngroups; h++) C1[(size_t)h * G + g + j] = two[(size_t)j * s.ngroups + h];
    }
    std::vector<u32> S1;
    excl_scan(C1, S1);
    u32 T = S1.back();
    std::vector<uint16_t> part_lo(T + 1);
    std::vector<u32> part_val(T + 1);
    const u32 cap = ((n + G - 1) / G) * s.nwin;
    std::vector<uint16_t> stage_lo(cap + 1), stage_grp(cap + 1);
    std::vector<u32> stage_val(cap + 1), hist(s.ngroups), loff;
    for (u32 g = 0; g < G; g++) {   // LDS-staged partition: count again, scan, place, copy
        for (u32 h = 0; h < s.ngroups; h++) hist[h] = 0;
        for (u32 t = 0; t < nthr; t++) msm2_count_body(s, (const Fr *)sc_v, mont != 0, g, hist.data(), t, nthr);
        excl_scan(hist, loff);
        if (loff[s.ngroups] > cap) return -1;
        for (u32 h = 0; h < s.ngroups; h++) hist[h] = loff[h];
#define EMU_PLACE(C) msm2_stage_place_body<C>(s, (const Fr *)sc_v, mont != 0, g, hist.data(), stage_lo.data(), stage_val.data(), stage_grp.data(), t, nthr)
        for (u32 t = 0; t < nthr; t++) { MSM2_FOR_C(s.c, EMU_PLACE) }
#undef EMU_PLACE
        std::vector<u32> gbase(s.ngroups);
        for (u32 h = 0; h < s.ngroups; h++) gbase[h] = S1[(size_t)h * G + g];
        for (u32 t = 0; t < nthr; t++) msm2_stage_copy_body(s, gbase.data(), loff.data(), stage_lo.data(), stage_val.data(), stage_grp.data(), part_lo.data(), part_val.data(), t, nthr);
    }
    std::vector<u32> gstart(s.ngroups + 1), cstart, nch(s.ngroups);
    for (u32 h = 0; h < s.ngroups; h++) msm2_chunk_count_body(s, S1.data(), gstart.data(), nch.data(), h);
    excl_scan(nch, cstart);
    u32 nchunks = cstart[s.ngroups];
    std::vector<u32> H2((size_t)(nchunks + 1) * s.gsize), total(s.nkeys);
    for (u32 ch = 0; ch < nchunks + 2; ch++) {   // over-launch like the bounded GPU grid
        u32 hi, b, e;
        if (!msm2_chunk_range(s, gstart.data(), cstart.data(), ch, hi, b, e)) continue;
        for (u32 t = 0; t < nthr; t++) msm2_hist2_zero(s, lds.data(), t, nthr);
        for (u32 t = 0; t < nthr; t++) msm2_hist2_count(part_lo.data(), b, e, lds.data(), t, nthr);
        for (u32 t = 0; t < nthr; t++) msm2_hist2_write(s, H2.data(), ch, lds.data(), t, nthr);
    }
    for (u32 k = 0; k < s.nkeys; k++) msm2_colsum_body(s, cstart.data(), H2.data(), total.data(), k);
    std::vector<u32> keystart;
    excl_scan(total, keystart);
    std::vector<u32> sorted(T + 1);
    for (u32 ch = 0; ch < nchunks; ch++) {
        u32 hi, b, e;
        msm2_chunk_range(s, gstart.data(), cstart.data(), ch, hi, b, e);
        for (u32 t = 0; t < nthr; t++) msm2_scatter2_init(s, keystart.data(), H2.data(), ch, hi, lds.data(), t, nthr);
        if (e - b <= MSM2_STAGE_PER * nthr && (ch & 1)) {   // every other chunk through the staged scatter (k_msm2_scatter2_staged's phases)
            std::vector<u32> loc(s.gsize, 0), cursor(lds.begin(), lds.begin() + s.gsize), st_val(e - b), rk((size_t)nthr * MSM2_STAGE_PER), vv((size_t)nthr * MSM2_STAGE_PER), mm(nthr);
            std::vector<uint16_t> st_lo(e - b), ll((size_t)nthr * MSM2_STAGE_PER);
            for (u32 t = 0; t < nthr; t++)
                mm[t] = msm2_stage2_rank(part_lo.data(), part_val.data(), b, e, loc.data(), t, nthr, &ll[(size_t)t * MSM2_STAGE_PER], &vv[(size_t)t * MSM2_STAGE_PER], &rk[(size_t)t * MSM2_STAGE_PER]);
            u32 run = 0;
            for (u32 q = 0; q < s.gsize; q++) { u32 c = loc[q]; loc[q] = run; run += c; }
            for (u32 t = 0; t < nthr; t++)
                msm2_stage2_place(loc.data(), &ll[(size_t)t * MSM2_STAGE_PER], &vv[(size_t)t * MSM2_STAGE_PER], &rk[(size_t)t * MSM2_STAGE_PER], mm[t], st_lo.data(), st_val.data());
            for (u32 t = 0; t < nthr; t++) msm2_stage2_copy(cursor.data(), loc.data(), st_lo.data(), st_val.data(), e - b, sorted.data(), t, nthr);
        } else
            for (u32 t = 0; t < nthr; t++) msm2_scatter2_move(part_lo.data(), part_val.data(), b, e, lds.data(), sorted.data(), t, nthr);
    }
    // items / levels over nkeys keys (one window), then bucket reduce with nwin = 1
    MsmShape ks;
    ks.c = c; ks.nwin = wkeys ? s.nwin : 1; ks.nbuckets = s.half; ks.nkeys = s.nkeys; ks.nslices = G; ks.n = n;
    std::vector<u32> start(s.nkeys), cnt(s.nkeys), items(s.nkeys);
    for (u32 k = 0; k < s.nkeys; k++) msm_prep_level1(ks, keystart.data(), L, start.data(), cnt.data(), items.data(), k);
    std::vector<XYZZ<F>> bucket(s.nkeys);
    memset(bucket.data(), 0, sizeof(XYZZ<F>) * s.nkeys);
    run_levels<F>(s.nkeys, start, cnt, items, L, pre.data(), sorted.data(), std::vector<XYZZ<F>>(), bucket, nthr);
    u32 tb = (s.half + seg - 1) / seg;
    std::vector<XYZZ<F>> Pp((size_t)ks.nwin * tb), wsum(ks.nwin);
    for (u32 w = 0; w < ks.nwin; w++) {
        for (u32 t = 0; t < tb; t++) msm_bucket_reduce_body<F>(bucket.data(), s.half, seg, Pp.data(), w, t);
        wsum[w] = XYZZ<F>::inf();
        for (u32 t = 0; t < tb; t++) xyzz_add(wsum[w], Pp[(size_t)w * tb + t]);
    }
    XYZZ<F> tot = msm_combine_windows<F>(wsum.data(), ks.nwin, wkeys ? c : 0);   // one window: no doublings
    *(Affine<F> *)out_v = xyzz_to_affine(tot);
    return (int)nchunks;
}
