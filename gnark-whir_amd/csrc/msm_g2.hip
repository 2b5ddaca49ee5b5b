// G2 instantiation of the MSM point kernels (see msm_curve_kernels.cuh, msm.hip) + the LDS-accumulator level-1 kernel.
#include "msm_curve_kernels.cuh"
#include "curve29_g2.cuh"

// G2 level-1 accumulation with the XYZZ accumulator resident in LDS ([word][lane] image, 16 KiB per 64-lane
// workgroup): only the operands of the current step live in VGPRs, so the kernel needs no scratch (the register
// version spilled 480 B per lane -- 9.5 GB of scratch writes per launch in the PMC pass).
struct LdsAccG2 {
    u32 *base;   // &lds[0][lane]
    MI_D Fp2 ld(int comp) const {
        Fp2 v;
#pragma unroll
        for (int i = 0; i < 8; i++) { v.a0.l[i] = base[(comp * 16 + i) * 64]; v.a1.l[i] = base[(comp * 16 + 8 + i) * 64]; }
        return v;
    }
    MI_D void st(int comp, const Fp2 &v) const {
#pragma unroll
        for (int i = 0; i < 8; i++) { base[(comp * 16 + i) * 64] = v.a0.l[i]; base[(comp * 16 + 8 + i) * 64] = v.a1.l[i]; }
    }
    MI_D G2X load() const { return G2X{ld(0), ld(1), ld(2), ld(3)}; }
    MI_D void store(const G2X &a) const { st(0, a.x); st(1, a.y); st(2, a.zz); st(3, a.zzz); }
};
// acc += (+/-) q  (madd-2008-s, same special cases as xyzz_madd); inf tracks "accumulator is the point at infinity"
MI_D void xyzz_madd_lds(const LdsAccG2 &A, bool &inf, const G2Aff &q, bool negate) {
    if (q.is_inf()) return;
    Fp2 qy = negate ? fe_neg(q.y) : q.y;
    if (inf) { A.st(0, q.x); A.st(1, qy); A.st(2, Fp2::one()); A.st(3, Fp2::one()); inf = false; return; }
    Fp2 U2 = q.x * A.ld(2);
    Fp2 S2 = qy * A.ld(3);
    Fp2 x = A.ld(0);
    Fp2 Pp = U2 - x;
    Fp2 R = S2 - A.ld(1);
    if (Pp.is_zero()) {   // rare: doubling or cancellation -> generic path through registers
        G2X acc = A.load();
        xyzz_madd(acc, q, negate);
        inf = acc.is_inf();
        A.store(acc);
        return;
    }
    Fp2 PP = fe_sqr(Pp);
    Fp2 PPP = Pp * PP;
    Fp2 Q = x * PP;
    A.st(2, A.ld(2) * PP);
    A.st(3, A.ld(3) * PPP);
    Fp2 X3 = fe_sqr(R) - PPP - fe_dbl(Q);
    A.st(0, X3);
    A.st(1, R * (Q - X3) - A.ld(1) * PPP);
}
__global__ void __launch_bounds__(64, 2) k_msm_accum_affine_g2_lds(const G2Aff *pts, const u32 *sorted, const u32 *start, const u32 *cnt,
                                                                   const u32 *items, const u32 *item_start, u32 nkeys, u32 L,
                                                                   G2X *bucket, G2X *partial_out) {
    __shared__ u32 lds[64 * 64];
    const LdsAccG2 A{&lds[threadIdx.x]};
    const u32 total = item_start[nkeys], stride = gridDim.x * blockDim.x;
    for (u32 item = blockIdx.x * blockDim.x + threadIdx.x; item < total; item += stride) {
        u32 key = msm_item_key(item_start, nkeys, item), b, e;
        msm_item_range(start[key], cnt[key], items[key], item - item_start[key], b, e);
        bool inf = true;
        for (u32 k = b; k < e; k++) {
            u32 v = sorted[k];
            xyzz_madd_lds(A, inf, pts[v & 0x7fffffffu], (v >> 31) != 0);
        }
        G2X acc = inf ? G2X::inf() : A.load();
        if (items[key] == 1) bucket[key] = acc; else partial_out[item] = acc;
    }
}

static void launch_accum_affine_g2(hipStream_t st, unsigned grid, const void *pts, const u32 *sorted, const u32 *start, const u32 *cnt, const u32 *items,
                                   const u32 *item_start, u32 nkeys, u32 L, void *bucket, void *pout) {
    hipLaunchKernelGGL(k_msm_accum_affine_g2_lds, dim3(grid), dim3(64), 0, st, (const G2Aff *)pts, sorted, start, cnt, items, item_start, nkeys, L,
                       (G2X *)bucket, (G2X *)pout);
}
// The same level-1 accumulation in the 9 x 29-bit representation (curve29_g2.cuh): accumulator image [word][lane] of 4 x 18 words
// (18 KiB per 64-lane workgroup), points read in the R' packed form, an item's sum converted back to the standard XYZZ once.
struct LdsAccG2_29 {
    u32 *base;   // &lds[0][lane]
    MI_D F2_29 ld(int comp) const {
        F2_29 v;
#pragma unroll
        for (int i = 0; i < 9; i++) { v.a0.l[i] = base[(comp * 18 + i) * 64]; v.a1.l[i] = base[(comp * 18 + 9 + i) * 64]; }
        return v;
    }
    MI_D void st(int comp, const F2_29 &v) const {
#pragma unroll
        for (int i = 0; i < 9; i++) { base[(comp * 18 + i) * 64] = v.a0.l[i]; base[(comp * 18 + 9 + i) * 64] = v.a1.l[i]; }
    }
};
// an accumulator (or the point at infinity) -> 64 packed words in the R' form (curve29_g2.cuh); infinity = all zero
static __device__ __forceinline__ void g2x29_store_rp(const LdsAccG2_29 &A, bool inf, G2X *dst) {
    uint4 *d4 = reinterpret_cast<uint4 *>(dst);
#pragma unroll
    for (int comp = 0; comp < 4; comp++) {
        u32 w[16];
        if (inf) {
#pragma unroll
            for (int i = 0; i < 16; i++) w[i] = 0;
        } else {
            f2_29_pack(A.ld(comp), w);
        }
#pragma unroll
        for (int q = 0; q < 4; q++) d4[4 * comp + q] = make_uint4(w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]);
    }
}
// WG = waves per workgroup (each wave has its own 18 KiB accumulator image)
template <int WG>
__global__ void __launch_bounds__(64 * WG, 2) k_msm_accum_affine_g2_29(const G2Aff *pts, const u32 *sorted, const uint4 *tab, const u32 *item_start, u32 nkeys,
                                                                       G2X *bucket, G2X *partial_out, u32 rp_partials) {
    __shared__ u32 lds[72 * 64 * WG];
    const LdsAccG2_29 A{&lds[(threadIdx.x >> 6) * (72 * 64) + (threadIdx.x & 63)]};
    const u32 total = item_start[nkeys], stride = gridDim.x * blockDim.x;
    for (u32 item = blockIdx.x * blockDim.x + threadIdx.x; item < total; item += stride) {
        const uint4 rec = tab[item];
        const u32 key = rec.x, b = rec.y, e = rec.z;
        bool inf = true;
        for (u32 k = b; k < e; k++) {
            const u32 v = sorted[k];
            const uint4 *q4 = reinterpret_cast<const uint4 *>(pts + (v & 0x7fffffffu));
            u32 w[32];
#pragma unroll
            for (int j = 0; j < 8; j++) { const uint4 t = q4[j]; w[4 * j] = t.x; w[4 * j + 1] = t.y; w[4 * j + 2] = t.z; w[4 * j + 3] = t.w; }
            g2x29_madd(A, inf, w, (v >> 31) != 0);
        }
        // a bucket's only item leaves in the standard form (what the bucket reduce reads); a partial sum stays in the R' form for the
        // next level (k_msm_accum_xyzz_g2_29): a pack instead of eight conversion products
        if (!rec.w && rp_partials) { g2x29_store_rp(A, inf, partial_out + item); continue; }
        G2X out = G2X::inf();
        if (!inf) out = G2X{f2_29_to_std(A.ld(0)), f2_29_to_std(A.ld(1)), f2_29_to_std(A.ld(2)), f2_29_to_std(A.ld(3))};
        if (rec.w) bucket[key] = out; else partial_out[item] = out;
    }
}
// Levels >= 2 of the item machinery over partial sums in the packed R' form: k_msm_accum_xyzz's decomposition, the additions in nine
// 29-bit limbs (g2x29_add) with the running sum in the same LDS image as level 1, the operand's coordinates read from global memory as
// each is needed; a bucket's final sum converted to the standard form once.
__global__ void __launch_bounds__(64, 2) k_msm_accum_xyzz_g2_29(const G2X *partial_in, const u32 *start, const u32 *cnt, const u32 *items, const u32 *item_start,
                                                                u32 nkeys, G2X *bucket, G2X *partial_out) {
    __shared__ u32 lds[72 * 64];
    const LdsAccG2_29 A{&lds[threadIdx.x]};
    const u32 total = item_start[nkeys], stride = gridDim.x * blockDim.x;
    for (u32 item = blockIdx.x * blockDim.x + threadIdx.x; item < total; item += stride) {
        const u32 key = msm_item_key(item_start, nkeys, item);
        u32 b, e;
        msm_item_range(start[key], cnt[key], items[key], item - item_start[key], b, e);
        bool inf = true;
        for (u32 k = b; k < e; k++) {
            const u32 *bw = reinterpret_cast<const u32 *>(partial_in + k);
            u32 any = 0;
#pragma unroll
            for (int i = 32; i < 48; i++) any |= bw[i];   // ZZ = 0 exactly: only the stored infinity
            g2x29_add(A, inf, [bw](int comp) { return f2_29_unpack(bw + 16 * comp); }, any == 0);
        }
        if (items[key] != 1) { g2x29_store_rp(A, inf, partial_out + item); continue; }
        G2X out = G2X::inf();
        if (!inf) out = G2X{f2_29_to_std(A.ld(0)), f2_29_to_std(A.ld(1)), f2_29_to_std(A.ld(2)), f2_29_to_std(A.ld(3))};
        bucket[key] = out;
    }
}
static void launch_accum_xyzz_g2_29(hipStream_t st, unsigned grid, const void *pin, const u32 *start, const u32 *cnt, const u32 *items, const u32 *item_start,
                                    u32 nkeys, u32 L, void *bucket, void *pout) {
    hipLaunchKernelGGL(k_msm_accum_xyzz_g2_29, dim3(grid), dim3(64), 0, st, (const G2X *)pin, start, cnt, items, item_start, nkeys, (G2X *)bucket, (G2X *)pout);
}
static void launch_accum_affine_g2_29(hipStream_t st, unsigned grid, const void *pts, const u32 *sorted, const u32 *start, const u32 *cnt, const u32 *items,
                                      const u32 *item_start, u32 nkeys, u32 L, void *bucket, void *pout, void *item_tab, u32 rp_partials,
                                      hipEvent_t ev_before) {
    hipLaunchKernelGGL(k_msm_item_table<Fp2>, dim3(grid < 32768 ? grid : 32768), dim3(64), 0, st, start, cnt, items, item_start, nkeys, (uint4 *)item_tab);
    if (ev_before) (void)hipEventRecord(ev_before, st);
    // rp_partials: bit 0 = partial sums stay in the R' form, bits 2..3 = log2 of the waves per workgroup (`grid` counts waves)
    const u32 wg = 1u << ((rp_partials >> 2) & 3u);
    const unsigned g = (grid + wg - 1) / wg;
#define MI_L1G2(WG) hipLaunchKernelGGL(k_msm_accum_affine_g2_29<WG>, dim3(g), dim3(64 * WG), 0, st, (const G2Aff *)pts, sorted, (const uint4 *)item_tab, item_start, nkeys, \
                                       (G2X *)bucket, (G2X *)pout, rp_partials & 1u)
    if (wg == 4) MI_L1G2(4); else if (wg == 2) MI_L1G2(2); else MI_L1G2(1);
#undef MI_L1G2
}
// the finisher over partial sums in the packed R' form: the running sum in the LDS image of the level kernels (k_msm_accum_xyzz_g2_29)
struct FinG2rp {
    typedef G2X Partial;
    static constexpr int LDS_WORDS_PER_WAVE = 72 * 64;
    LdsAccG2_29 A;
    bool inf = true;
    MI_D explicit FinG2rp(u32 *lane_base) : A{lane_base} {}
    MI_D void add(const Partial *p) {
        const u32 *bw = reinterpret_cast<const u32 *>(p);
        u32 any = 0;
#pragma unroll
        for (int i = 32; i < 48; i++) any |= bw[i];   // ZZ = 0 exactly: only the stored infinity
        g2x29_add(A, inf, [bw](int comp) { return f2_29_unpack(bw + 16 * comp); }, any == 0);
    }
    MI_D void load(const Partial *p) { inf = true; add(p); }   // (infinity + b = b: four component copies into the image)
    MI_D void store(Partial *p) const { g2x29_store_rp(A, inf, p); }
    MI_D void to_bucket(G2X *b) const {
        G2X out = G2X::inf();
        if (!inf) out = G2X{f2_29_to_std(A.ld(0)), f2_29_to_std(A.ld(1)), f2_29_to_std(A.ld(2)), f2_29_to_std(A.ld(3))};
        *b = out;
    }
};
static void launch_finish_g2(hipStream_t st, unsigned nb_small, unsigned nb_big, void *partials, const u32 *list_small, const u32 *list_big, const u32 *counters,
                             const u32 *item_start, const u32 *items, void *bucket, u32 rp) {
    if (rp) launch_finish_form<FinG2rp, G2X, 64, 2>(st, nb_small, nb_big, partials, list_small, list_big, counters, item_start, items, bucket);
    else launch_finish_form<FinStd<Fp2>, G2X, 64, 2>(st, nb_small, nb_big, partials, list_small, list_big, counters, item_start, items, bucket);
}
__global__ void k_g2_to_rprime(G2Aff *dst, const G2Aff *src, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    G2Aff a = src[i];
    dst[i] = G2Aff{Fp2{fe_to_rprime_packed(a.x.a0), fe_to_rprime_packed(a.x.a1)}, Fp2{fe_to_rprime_packed(a.y.a0), fe_to_rprime_packed(a.y.a1)}};
}
static void launch_g2_to_rprime(hipStream_t st, void *dst, const void *src, size_t n) {
    if (n) hipLaunchKernelGGL(k_g2_to_rprime, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (G2Aff *)dst, (const G2Aff *)src, n);
}

const MsmCurveOps &msm_g2_ops() {
    static const MsmCurveOps ops = {sizeof(G2X), launch_accum_affine_g2, launch_accum_xyzz<Fp2>, launch_bucket_reduce<Fp2>, SumT<Fp2>::value, launch_sum_tree<Fp2>, launch_precompute<Fp2>, launch_precompute_batched<Fp2>, sizeof(Fp2), host_combine_windows<Fp2>, launch_sum_slices<Fp2>, launch_accum_affine_g2_29, launch_accum_xyzz_g2_29, nullptr, launch_finish_g2, 64, 1024, launch_g2_to_rprime};
    return ops;
}
