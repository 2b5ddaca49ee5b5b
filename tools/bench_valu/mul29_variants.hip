// Probe: variants of the 9 x 29-bit Montgomery product (field29.cuh) as dependent chains on all CUs, at 8 and at 3 waves per SIMD.
//   C      the production source form (the compiler splits every column into a fresh accumulator + a 64-bit add)
//   ASM    every multiply-accumulate an opaque v_mad_u64_u32 into ONE accumulator chain (no 64-bit adds)
//   SQR    ASM with the 45-product squaring
// hipcc -O3 --offload-arch=gfx950 mul29_variants.hip -o mul29_variants && ./mul29_variants
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32;
typedef uint64_t u64;
struct F29 { u32 l[9]; };
#define P0 0x187cfd47u
#define P1 0x10460b6u
#define P2 0x1c72a34fu
#define P3 0x2d522d0u
#define P4 0x1585d978u
#define P5 0x2db40c0u
#define P6 0xa6e141u
#define P7 0xe5c2634u
#define P8 0x30644eu
__device__ __constant__ const u32 PP[9] = {P0, P1, P2, P3, P4, P5, P6, P7, P8};

template <int V> __device__ __forceinline__ void mac(u64 &acc, u32 a, u32 b) {
    if (V == 0) acc += (u64)a * b;
    else { u64 c; asm("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(acc), "=s"(c) : "v"(a), "v"(b)); }
}
template <int V> __device__ __forceinline__ void macs(u64 &acc, u32 a, u32 s) {   // s: wave-uniform constant
    if (V == 0) acc += (u64)a * s;
    else { u64 c; asm("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(acc), "=s"(c) : "v"(a), "s"(s)); }
}
template <int V, bool SQ> __device__ __forceinline__ F29 mul29(const F29 &x, const F29 &y, u32 inv) {
    const u32 M = (1u << 29) - 1;
    const u32 p[9] = {P0, P1, P2, P3, P4, P5, P6, P7, P8};
    u32 m[9], x2[9];
    if (SQ) {
#pragma unroll
        for (int i = 0; i < 9; i++) x2[i] = x.l[i] << 1;
    }
    F29 r;
    u64 acc = 0;
#pragma unroll
    for (int k = 0; k < 17; k++) {
        const int lo = k < 9 ? 0 : k - 8, hi = k < 9 ? k : 8;
        if (SQ) {
#pragma unroll
            for (int i = lo; i <= hi; i++) { const int j = k - i; if (i < j) mac<V>(acc, x2[i], x.l[j]); else if (i == j) mac<V>(acc, x.l[i], x.l[i]); }
        } else {
#pragma unroll
            for (int i = lo; i <= hi; i++) mac<V>(acc, x.l[i], y.l[k - i]);
        }
#pragma unroll
        for (int i = lo; i <= hi; i++) if (k - i > 0 || k >= 9) { if (k < 9 ? i < k : true) macs<V>(acc, m[i], p[k - i]); }
        if (k < 9) { m[k] = ((u32)acc * inv) & M; macs<V>(acc, m[k], p[0]); }
        else r.l[k - 9] = (u32)acc & M;
        acc >>= 29;
    }
    r.l[8] = (u32)acc;
    return r;
}
template <int V, bool SQ>
__global__ void __launch_bounds__(256) kk(F29 *out, u32 iters, u32 inv) {
    extern __shared__ u32 lds[];
    F29 a, b;
#pragma unroll
    for (int i = 0; i < 9; i++) { a.l[i] = (threadIdx.x * 2654435761u + i * 40503u) & ((1u << 28) - 1); b.l[i] = (blockIdx.x * 2246822519u + i * 7919u) & ((1u << 28) - 1); }
    for (u32 it = 0; it < iters; it++) { a = mul29<V, SQ>(a, b, inv); b = mul29<V, SQ>(b, a, inv); }
    if (a.l[0] == 0xdeadbeef) { out[0] = b; lds[threadIdx.x] = 1; }
    if (blockIdx.x == 0) out[threadIdx.x] = a;   // one writer per slot: the signature below is deterministic
}
// host check of the variants against each other is left to the production tests; here the three kernels must agree on out[]
template <int V, bool SQ> static void run(const char *name, u32 inv, F29 *buf, size_t lds, u32 *sig) {
    const u32 blocks = 4096, threads = 256, iters = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 1e9;
    for (int rep = 0; rep < 3; rep++) {
        float t;
        hipEventRecord(e0); hipLaunchKernelGGL((kk<V, SQ>), dim3(blocks), dim3(threads), lds, 0, buf, iters, inv); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&t, e0, e1);
        if (t < ms) ms = t;
    }
    F29 h[4]; hipMemcpy(h, buf, sizeof(h), hipMemcpyDeviceToHost);
    *sig = h[1].l[0] ^ h[2].l[3] ^ h[3].l[8];
    printf("%-6s lds %6zu B: %.3f ms = %.1f G products/s   (sig %08x)\n", name, lds, ms, (double)blocks * threads * iters * 2 / ms / 1e6, *sig);
}
int main() {
    u32 inv = 1;
    for (int i = 0; i < 6; i++) inv *= 2 - P0 * inv;
    inv = (0u - inv) & ((1u << 29) - 1);
    F29 *buf; hipMalloc(&buf, 1024 * sizeof(F29));
    u32 s0, s1, s2;
    for (size_t lds : {(size_t)0, (size_t)50 * 1024}) {   // 50 KB per 4-wave workgroup: 3 workgroups per CU = 3 waves per SIMD
        run<0, false>("C", inv, buf, lds, &s0);
        run<1, false>("ASM", inv, buf, lds, &s1);
        run<1, true>("SQR", inv, buf, lds, &s2);
        if (s0 != s1) printf("MISMATCH C vs ASM\n");
    }
    return 0;
}
