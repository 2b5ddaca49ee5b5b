// G2 instantiation of the MSM point kernels (see msm_curve_kernels.cuh, msm.hip) + the LDS-accumulator level-1 kernel.
#include "msm_curve_kernels.cuh"

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
const MsmCurveOps &msm_g2_ops() {
    static const MsmCurveOps ops = {sizeof(G2X), launch_accum_affine_g2, launch_accum_xyzz<Fp2>, launch_bucket_reduce<Fp2>, SumT<Fp2>::value, launch_sum_tree<Fp2>, launch_precompute<Fp2>, host_combine_windows<Fp2>, launch_sum_slices<Fp2>, nullptr, nullptr};
    return ops;
}
