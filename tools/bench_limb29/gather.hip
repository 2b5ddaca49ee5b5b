// Probe: how many random 64-byte gathers per second does the memory system deliver (no arithmetic)?  Table of 2^LOG points of 64 B.
// hipcc -O3 --offload-arch=gfx950 gather.hip -o gather && LOG=27 ./gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
__global__ void __launch_bounds__(64) kgather(const uint4 *tab, uint64_t mask, uint32_t iters, uint4 *out, int dep) {
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint4 acc = make_uint4(tid, 0, 0, 0);
    uint64_t h = tid * 0x9E3779B97F4A7C15ull;
    for (uint32_t it = 0; it < iters; it++) {
        h = h * 6364136223846793005ull + 1442695040888963407ull + (dep ? acc.x : 0);   // dep: the next address needs this gather (a chain)
        const uint4 *q = tab + ((h >> 20) & mask) * 4;
        uint4 a = q[0], b = q[1], c = q[2], d = q[3];
        acc.x ^= a.x ^ b.y ^ c.z ^ d.w; acc.y += a.y + d.x; acc.z ^= b.z; acc.w += c.w;
    }
    out[tid & 1023] = acc;
}
int main() {
    const int lg = getenv("LOG") ? atoi(getenv("LOG")) : 27;
    const size_t n = (size_t)1 << lg;
    uint4 *tab, *out;
    if (hipMalloc(&tab, n * 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMalloc(&out, 1024 * 16);
    hipMemset(tab, 1, n * 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int waves = 1; waves <= 8; waves *= 2) for (int dep = 0; dep < 2; dep++) {
        const uint32_t threads = 256 * 4 * 64 * waves, iters = 256;   // `waves` waves per SIMD resident
        float ms;
        hipLaunchKernelGGL(kgather, dim3(threads / 64), dim3(64), 0, 0, tab, n - 1, 8, out, dep);
        hipEventRecord(e0); hipLaunchKernelGGL(kgather, dim3(threads / 64), dim3(64), 0, 0, tab, n - 1, iters, out, dep); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("table %.1f GB, %d waves/SIMD, %s: %.2f G gathers/s = %.0f GB/s of useful bytes\n", n * 64 / 1e9, waves, dep ? "dependent chain" : "independent", (double)threads * iters / ms / 1e6,
               (double)threads * iters * 64 / ms / 1e6);
    }
    return 0;
}
