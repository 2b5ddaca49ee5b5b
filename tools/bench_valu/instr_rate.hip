// Probe: issue rate of the integer instructions the 256-bit modular products are made of, per SIMD, on gfx950.
// Each kernel runs UNROLL independent chains of one instruction (8 waves per SIMD resident): the time per instruction is its
// issue cost; full rate = 4 cycles per wave64 instruction.   hipcc -O3 --offload-arch=gfx950 instr_rate.hip -o instr_rate && ./instr_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32;
typedef uint64_t u64;
#define CHAINS 8
template <int OP>
__global__ void __launch_bounds__(256) k(u32 *out, u32 iters, u32 seed) {
    u64 a[CHAINS];
    u32 b = seed + threadIdx.x, c = seed * 3 + 1;
    for (int i = 0; i < CHAINS; i++) a[i] = (u64)(threadIdx.x + i) * 0x9E3779B97F4A7C15ull;
    for (u32 it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
#pragma unroll
            for (int i = 0; i < CHAINS; i++) {
                u32 lo = (u32)a[i], hi = (u32)(a[i] >> 32);
                if (OP == 0) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c) : "vcc");
                if (OP == 1) asm volatile("v_lshrrev_b64 %0, 29, %0" : "+v"(a[i]));
                if (OP == 2) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(a[i]) : "v"(a[(i + 1) % CHAINS]));
                if (OP == 3) { asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(lo) : "v"(c)); a[i] = ((u64)hi << 32) | lo; }
                if (OP == 4) { asm volatile("v_add_u32 %0, %0, %1" : "+v"(lo) : "v"(c)); a[i] = ((u64)hi << 32) | lo; }
                if (OP == 5) { asm volatile("v_alignbit_b32 %0, %1, %0, 29" : "+v"(lo) : "v"(hi)); a[i] = ((u64)hi << 32) | lo; }
                if (OP == 6) { asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(lo), "+v"(hi) : "v"(b), "v"(c) : "vcc"); a[i] = ((u64)hi << 32) | lo; }
                if (OP == 7) { asm volatile("v_and_b32 %0, 0x1fffffff, %0" : "+v"(lo)); a[i] = ((u64)hi << 32) | lo; }
                if (OP == 8) { asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(lo) : "v"(c)); a[i] = ((u64)hi << 32) | lo; }
                if (OP == 9) { asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(lo) : "v"(c), "v"(b)); a[i] = ((u64)hi << 32) | lo; }
                if (OP == 10) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[i]) : "v"(b), "s"(c) : "vcc");
                if (OP == 11) { asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(lo) : "v"(c), "v"(b)); a[i] = ((u64)hi << 32) | lo; }
                if (OP == 12) asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(a[i]));
                if (OP == 13) { asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(lo) : "v"(c)); a[i] = ((u64)hi << 32) | lo; }
                if (OP == 14) { asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(lo) : "v"(c)); a[i] = ((u64)hi << 32) | lo; }
                if (OP == 15) { asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(lo) : "v"(c)); a[i] = ((u64)hi << 32) | lo; }
            }
        }
    }
    u64 s = 0;
    for (int i = 0; i < CHAINS; i++) s ^= a[i];
    out[(blockIdx.x * blockDim.x + threadIdx.x) & 1023] = (u32)s ^ (u32)(s >> 32);
}
template <int OP>
static void run(const char *name, int per) {
    u32 *out; hipMalloc(&out, 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const u32 iters = 2000, blocks = 256 * 8;   // 8 workgroups of 4 waves per CU = 8 waves per SIMD
    float ms;
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 10, 7);
    hipEventRecord(e0); hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, 7); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    const double wave_instrs = (double)blocks * 4 * iters * 16 * CHAINS * per;
    const double per_simd_per_s = wave_instrs / (ms * 1e-3) / 1024.0;
    printf("%-28s %8.3f ms  %.3e wave-instr/s/SIMD  = %.2f cycles per instruction at 2.4 GHz\n", name, ms, per_simd_per_s, 2.4e9 / per_simd_per_s);
    hipFree(out);
}
int main() {
    run<0>("v_mad_u64_u32 (v,v)", 1); run<10>("v_mad_u64_u32 (v,s)", 1); run<1>("v_lshrrev_b64", 1); run<2>("v_lshl_add_u64", 1);
    run<3>("v_mul_lo_u32", 1); run<8>("v_mul_hi_u32", 1); run<4>("v_add_u32", 1); run<11>("v_add3_u32", 1); run<5>("v_alignbit_b32", 1);
    run<6>("v_add_co + v_addc_co", 2); run<7>("v_and_b32", 1); run<9>("v_mad_u32_u24", 1); run<13>("v_lshl_add_u32", 1);
    run<14>("v_mul_u32_u24", 1); run<15>("v_mul_hi_u32_u24", 1); run<12>("v_fma_f64", 1);
    return 0;
}
