// Probe: G1 XYZZ mixed addition in the 9 x 29-bit representation (curve29.cuh) against the production one (curve.cuh):
// same points, results compared, chains of additions per thread on all CUs.
// hipcc -O3 -std=c++17 --offload-arch=gfx950 -I../../gnark-whir_amd/csrc madd29.hip -o madd29 && ./madd29
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "curve29.cuh"

__global__ void k_gen(G1Aff *pts, G1Aff *pts29, u32 n) {   // k * G by repeated addition is too slow: points (x, sqrt(x^3+3)) by trial
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fp three = fe_from_u32<FpParams>(3);
    u32 e[8] = {0xb61f3f52u, 0x4f082305u, 0x5a1c72a3u, 0x65e05aa4u, 0xa0605617u, 0x6e14116du, 0xb84c680au, 0x0c19139cu};   // (p + 1) / 4
    pts[i] = G1Aff{Fp::zero(), Fp::zero()}; pts29[i] = pts[i];   // stays infinity if no trial succeeds (bounded loop)
    for (u32 t = 0; t < 64; t++) {
        Fp x = fe_from_u32<FpParams>(1000003u * i + 7919u * t + 12345u);
        Fp rhs = x * x * x + three;
        Fp y = fe_pow(rhs, e);
        if (y * y == rhs) {
            pts[i] = G1Aff{x, y};
            pts29[i] = G1Aff{fe_to_rprime_packed(x), fe_to_rprime_packed(y)};
            return;
        }
    }
}
__global__ void __launch_bounds__(64) k32(const G1Aff *pts, u32 mask, u32 iters, G1X *out) {
    u32 tid = blockIdx.x * blockDim.x + threadIdx.x;
    G1X acc = G1X::inf();
    for (u32 it = 0; it < iters; it++) {
        u32 idx = ((tid * 2654435761u) ^ (it * 40503u + (tid >> 3) * 2246822519u)) & mask;
        xyzz_madd(acc, pts[idx], (it & 3) == 3);
    }
    out[tid] = acc;
}
template <int W>
__global__ void __launch_bounds__(64, W) k29(const G1Aff *pts29, u32 mask, u32 iters, G1X *out) {
    u32 tid = blockIdx.x * blockDim.x + threadIdx.x;
    G1X29 acc = g1x29_inf();
    for (u32 it = 0; it < iters; it++) {
        u32 idx = ((tid * 2654435761u) ^ (it * 40503u + (tid >> 3) * 2246822519u)) & mask;
        const u32 *q = (const u32 *)&pts29[idx];
        u32 w[16];
#pragma unroll
        for (int i = 0; i < 16; i++) w[i] = q[i];
        g1x29_madd(acc, w, (it & 3) == 3);
    }
    out[tid] = g1x29_to_std(acc);
}
int main() {
    const u32 npts = getenv("NPTS_LOG") ? 1u << atoi(getenv("NPTS_LOG")) : 4096, threads = 256 * 64 * 8, iters = getenv("ITERS") ? atoi(getenv("ITERS")) : 64;
    G1Aff *pts, *pts29;
    G1X *o32, *o29;
    hipMalloc(&pts, npts * sizeof(G1Aff)); hipMalloc(&pts29, npts * sizeof(G1Aff));
    hipMalloc(&o32, threads * sizeof(G1X)); hipMalloc(&o29, threads * sizeof(G1X));
    // a big table repeats 65536 distinct points (the generator is slow); what matters is WHERE the 64 bytes come from
    hipLaunchKernelGGL(k_gen, dim3((npts < 65536 ? npts : 65536) / 64), dim3(64), 0, 0, pts, pts29, npts < 65536 ? npts : 65536);
    hipDeviceSynchronize();
    for (size_t off = 65536; off < npts; off += 65536) { hipMemcpy(pts + off, pts, 65536 * sizeof(G1Aff), hipMemcpyDeviceToDevice); hipMemcpy(pts29 + off, pts29, 65536 * sizeof(G1Aff), hipMemcpyDeviceToDevice); }
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; rep++) {
        float a, b;
        hipEventRecord(e0); hipLaunchKernelGGL(k32, dim3(threads / 64), dim3(64), 0, 0, pts, npts - 1, iters, o32); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&a, e0, e1);
        hipEventRecord(e0); hipLaunchKernelGGL(k29<1>, dim3(threads / 64), dim3(64), 0, 0, pts29, npts - 1, iters, o29); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&b, e0, e1);
        float c3, c4;
        hipEventRecord(e0); hipLaunchKernelGGL(k29<3>, dim3(threads / 64), dim3(64), 0, 0, pts29, npts - 1, iters, o29); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&c3, e0, e1);
        hipEventRecord(e0); hipLaunchKernelGGL(k29<4>, dim3(threads / 64), dim3(64), 0, 0, pts29, npts - 1, iters, o29); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&c4, e0, e1);
        printf("   9x29 at 3 waves/SIMD: %.3f ms = %.2f G/s | at 4 waves/SIMD: %.3f ms = %.2f G/s\n", c3, (double)threads * iters / c3 / 1e6, c4, (double)threads * iters / c4 / 1e6);
        const double n = (double)threads * iters;
        printf("8x32: %.3f ms = %.2f G madd/s | 9x29: %.3f ms = %.2f G madd/s (conversion of the result included)\n", a, n / a / 1e6, b, n / b / 1e6);
    }
    // same points (as group elements)?  compare the affine forms
    std::vector<G1X> h32(4096), h29(4096);
    hipMemcpy(h32.data(), o32, 4096 * sizeof(G1X), hipMemcpyDeviceToHost);
    hipMemcpy(h29.data(), o29, 4096 * sizeof(G1X), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 4096; i++) {
        G1Aff a = xyzz_to_affine(h32[i]), b = xyzz_to_affine(h29[i]);
        if (!(a.x == b.x) || !(a.y == b.y)) bad++;
    }
    printf("mismatching results: %d of 4096\n", bad);
    return bad != 0;
}
