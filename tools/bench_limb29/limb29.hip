// Probe: 9 x 29-bit unsaturated-limb Montgomery product (R = 2^261, no carry words: a column of <= 18 products of < 2^58 fits a
// 64-bit accumulator) against the production 8 x 32-bit product (field.cuh), as dependent chains on all CUs.
// hipcc -O3 --offload-arch=gfx950 -I../../gnark-whir_amd/csrc limb29.hip -o limb29 && ./limb29
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "field.cuh"

// BN254 Fp modulus in 9 x 29-bit limbs, and -p^-1 mod 2^29
__constant__ uint32_t P29[9];
__constant__ uint32_t INV29;
struct F29 { uint32_t l[9]; };

__device__ __forceinline__ F29 mul29(const F29 &x, const F29 &y, const uint32_t *p, uint32_t inv) {
    const uint32_t M = (1u << 29) - 1;
    uint32_t m[9];
    F29 r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) acc += (uint64_t)x.l[i] * y.l[k - i];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * p[k - i];
        m[k] = ((uint32_t)acc * inv) & M;
        acc += (uint64_t)m[k] * p[0];
        acc >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 17; k++) {
#pragma unroll
        for (int i = k - 8; i < 9; i++) acc += (uint64_t)x.l[i] * y.l[k - i];
#pragma unroll
        for (int i = k - 8; i < 9; i++) acc += (uint64_t)m[i] * p[k - i];
        r.l[k - 9] = (uint32_t)acc & M;
        acc >>= 29;
    }
    r.l[8] = (uint32_t)acc;
    return r;   // < 2p for operands < 2p (R = 2^261 > 64 p): lazy, no final subtraction
}

__global__ void k29(F29 *out, uint32_t iters) {
    uint32_t p[9];
#pragma unroll
    for (int i = 0; i < 9; i++) p[i] = P29[i];
    const uint32_t inv = INV29;
    F29 a, b;
#pragma unroll
    for (int i = 0; i < 9; i++) { a.l[i] = (threadIdx.x * 2654435761u + i * 40503u) & ((1u << 28) - 1); b.l[i] = (blockIdx.x * 2246822519u + i * 7919u) & ((1u << 28) - 1); }
    for (uint32_t it = 0; it < iters; it++) { a = mul29(a, b, p, inv); b = mul29(b, a, p, inv); }
    if (a.l[0] == 0xdeadbeef) out[0] = b;   // keep the chain alive
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x & 1023] = a;
}
__global__ void k32(Fp *out, uint32_t iters) {
    Fp a, b;
#pragma unroll
    for (int i = 0; i < 8; i++) { a.l[i] = threadIdx.x * 2654435761u + i * 40503u; b.l[i] = blockIdx.x * 2246822519u + i * 7919u; }
    a.l[7] &= 0x0fffffff; b.l[7] &= 0x0fffffff;
    for (uint32_t it = 0; it < iters; it++) { a = a * b; b = b * a; }
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x & 1023] = a + b;
}

int main() {
    // p = 21888242871839275222246405745257275088696311157297823662689037894645226208583
    const uint32_t p32[8] = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    uint32_t p29[9];
    for (int i = 0; i < 9; i++) {
        int bit = 29 * i, w = bit >> 5, sh = bit & 31;
        uint64_t v = p32[w] >> sh;
        if (w + 1 < 8) v |= (uint64_t)p32[w + 1] << (32 - sh);
        p29[i] = (uint32_t)(v & ((1u << 29) - 1));
    }
    uint32_t inv = 1;
    for (int i = 0; i < 6; i++) inv *= 2 - p29[0] * inv;
    inv = (0u - inv) & ((1u << 29) - 1);
    hipMemcpyToSymbol(HIP_SYMBOL(P29), p29, sizeof(p29));
    hipMemcpyToSymbol(HIP_SYMBOL(INV29), &inv, 4);
    void *buf;
    hipMalloc(&buf, 1024 * 64);
    const uint32_t blocks = 4096, threads = 256, iters = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; rep++) {
        float ms29, ms32;
        hipEventRecord(e0); hipLaunchKernelGGL(k29, dim3(blocks), dim3(threads), 0, 0, (F29 *)buf, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms29, e0, e1);
        hipEventRecord(e0); hipLaunchKernelGGL(k32, dim3(blocks), dim3(threads), 0, 0, (Fp *)buf, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms32, e0, e1);
        const double n = (double)blocks * threads * iters * 2;
        printf("9x29 lazy: %.3f ms = %.1f G products/s | 8x32 production: %.3f ms = %.1f G products/s\n", ms29, n / ms29 / 1e6, ms32, n / ms32 / 1e6);
    }
    return 0;
}
