// G1 instantiation of the MSM point kernels (see msm_curve_kernels.cuh, msm.hip).
#include "msm_curve_kernels.cuh"

const MsmCurveOps &msm_g1_ops() {
    static const MsmCurveOps ops = {sizeof(G1X), launch_accum_affine<Fp>, launch_accum_xyzz<Fp>, launch_bucket_reduce<Fp>, SumT<Fp>::value, launch_sum_tree<Fp>, launch_precompute<Fp>, host_combine_windows<Fp>, launch_sum_slices<Fp>};
    return ops;
}
