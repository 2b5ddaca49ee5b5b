// Probe (VERDICT r2 "next" item 7): batch x CONSTANT 256-bit products on the matrix cores of gfx950.
//
// Every VALU formulation of the 256-bit modular product has been measured (32-bit, 29-bit, fp64) while the matrix
// cores sit idle.  A product of two VARIABLE operands has no matrix shape (each element would need its own Toeplitz matrix), but a
// product by a CONSTANT over a batch is a small integer GEMM:  T = c * y  <=>  col[k] = sum_j  t[k - j] * s[j]  = (Toeplitz(t) x S)
// with the constant's digits t as the A matrix (64 x 32, shared by the whole batch) and one batch element's digits s per column of B.
// The constants in question: the Montgomery steps  m = T_lo * p' mod 2^256  and  m * p,  and the NTT's twiddles.
//
// What this measures, per wave of 64 elements (one per lane), with v_mfma_i32_32x32x32_i8 (signed 8-bit digits):
//   digits     y's bytes minus 128 (one v_xor per word; the constant offset c * 128 * (256^32 - 1) / 255 rides in the MFMA's C input);
//              c recoded once on the host into 33 signed digits
//   operands   B[k][column]: lane (r, h) supplies bytes 16h .. 16h+15 of element r -- half of them come from lane ^ 32 (ds_bpermute)
//   4 MFMAs    2 row blocks (64 byte columns of the 512-bit result) x 2 column groups (elements 0..31, 32..63)
//   carries    the i32 column sums (|.| < 2^20) go back to sixteen 32-bit words: four columns -> one 64-bit partial word, the two lane
//              halves exchange the words they hold of each other's element, one carry chain over the 16 words
// against the same 256 x 256 -> 512-bit constant product on the VALU (v_mad_u64_u32 product scanning, constant in scalar registers),
// both checked against host big-integer arithmetic on the first elements.
//   hipcc -O3 --offload-arch=gfx950 constmul.hip -o constmul && ./constmul
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
typedef uint32_t u32;
typedef uint64_t u64;
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

struct MfmaConst {
    v4i afrag[2][64];     // A fragment of row block b for lane l: A[row r][k = 16h + j] = t[32b + r - 16h - j]
    int cinit[2][2][16];  // C input of row block b, lane half h, register reg: 128 * sum of the digits t that meet column 32b + row
};

__device__ __forceinline__ int bperm(u32 src_lane, int v) { return __builtin_amdgcn_ds_bpermute((int)(src_lane << 2), v); }

// T = c * y (512 bits, 16 words) for the lane's element y (8 words), on the matrix cores
__device__ __forceinline__ void constmul_mfma(const MfmaConst *mc, const u32 *y, u32 *T, u32 lane) {
    const u32 h = lane >> 5, other = lane ^ 32;
    int s[8];
#pragma unroll
    for (int i = 0; i < 8; i++) s[i] = (int)(y[i] ^ 0x80808080u);
    // h = 0 lanes hold element r: they keep bytes 0..15 for column group 0 and send bytes 16..31 to lane r + 32 (the k = 16..31 half of
    // the same column); h = 1 lanes hold element 32 + r: they keep bytes 16..31 for group 1 and send bytes 0..15 to lane r
    v4i b0, b1;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int recv = bperm(other, h ? s[i] : s[4 + i]);
        b0[i] = h ? recv : s[i];
        b1[i] = h ? s[4 + i] : recv;
    }
    v16i d[2][2];
#pragma unroll
    for (int b = 0; b < 2; b++) {
        v16i c0;
#pragma unroll
        for (int r = 0; r < 16; r++) c0[r] = mc->cinit[b][h][r];
        const v4i a = mc->afrag[b][lane];
        d[b][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b0, c0, 0, 0, 0);
        d[b][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b1, c0, 0, 0, 0);
    }
    // registers 4q .. 4q+3 of (block b, group g) are the byte columns 32b + 8q + 4h + {0..3} of element 32g + r: one 64-bit partial word
    // (index 8b + 2q + h).  A lane keeps the words of its own element (group h) and trades those of the other group with lane ^ 32.
    int64_t own[2][4], got[2][4];
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            int64_t w[2];
#pragma unroll
            for (int g = 0; g < 2; g++) {
                const int lo = d[b][g][4 * q] + (d[b][g][4 * q + 1] << 8), hi = d[b][g][4 * q + 2] + (d[b][g][4 * q + 3] << 8);   // |.| < 2^29
                w[g] = (int64_t)lo + ((int64_t)hi << 16);
            }
            const int64_t mine = h ? w[1] : w[0], send = h ? w[0] : w[1];
            own[b][q] = mine;
            const u32 rl = (u32)bperm(other, (int)(u32)send), rh = (u32)bperm(other, (int)(u32)((u64)send >> 32));
            got[b][q] = (int64_t)(((u64)rh << 32) | rl);
        }
    // words in order: index 8b + 2q is held by the h = 0 lane of the pair, 8b + 2q + 1 by the h = 1 lane
    int64_t carry = 0;
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int64_t even = h ? got[b][q] : own[b][q], odd = h ? own[b][q] : got[b][q];
            int64_t t = even + carry;
            T[8 * b + 2 * q] = (u32)t; carry = t >> 32;
            t = odd + carry;
            T[8 * b + 2 * q + 1] = (u32)t; carry = t >> 32;
        }
}

// the same product on the VALU: product scanning, 96-bit column accumulator, the constant in scalar registers
__device__ __forceinline__ void constmul_valu(const u32 *c, const u32 *y, u32 *T) {
    u64 acc = 0;
    u32 top = 0;
#pragma unroll
    for (int k = 0; k < 15; k++) {
#pragma unroll
        for (int i = (k < 8 ? 0 : k - 7); i <= (k < 8 ? k : 7); i++) {
            const u64 p = (u64)c[i] * y[k - i];
            acc += p;
            top += acc < p ? 1u : 0u;
        }
        T[k] = (u32)acc;
        acc = (acc >> 32) | ((u64)top << 32);
        top = 0;
    }
    T[15] = (u32)acc;
}

template <int MODE>
__global__ void __launch_bounds__(256) k_chain(const MfmaConst *mc, const u32 *cw, const u32 *in, u32 *out, u32 iters) {
    const u32 lane = threadIdx.x & 63;
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    u32 y[8], T[16], c[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { y[i] = in[e * 8 + i]; c[i] = __builtin_amdgcn_readfirstlane(cw[i]); }
    for (u32 it = 0; it < iters; it++) {
        if (MODE == 0) constmul_mfma(mc, y, T, lane); else constmul_valu(c, y, T);
        if (it + 1 < iters)
#pragma unroll
            for (int i = 0; i < 8; i++) y[i] = T[i] ^ T[8 + i];   // dependent chain
    }
#pragma unroll
    for (int i = 0; i < 16; i++) out[e * 16 + i] = T[i];
}

static void host_mul(const u32 *c, const u32 *y, u32 *T) {
    u64 col[17] = {0};
    std::memset(T, 0, 64);
    unsigned __int128 acc = 0;
    for (int k = 0; k < 16; k++) {
        for (int i = 0; i < 8; i++) { int j = k - i; if (j >= 0 && j < 8) acc += (unsigned __int128)c[i] * y[j]; }
        T[k] = (u32)acc; acc >>= 32;
    }
    (void)col;
}

int main() {
    const u32 c[8] = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};   // BN254 p
    MfmaConst mc;
    int t[33];
    {
        int carry = 0;
        for (int i = 0; i < 32; i++) {
            int v = (int)((c[i / 4] >> (8 * (i % 4))) & 255) + carry;
            if (v >= 128) { t[i] = v - 256; carry = 1; } else { t[i] = v; carry = 0; }
        }
        t[32] = carry;
    }
    for (int b = 0; b < 2; b++)
        for (int l = 0; l < 64; l++) {
            const int r = l & 31, h = l >> 5;
            signed char byte[16];
            for (int j = 0; j < 16; j++) { const int idx = 32 * b + r - 16 * h - j; byte[j] = (idx >= 0 && idx <= 32) ? (signed char)t[idx] : 0; }
            std::memcpy(&mc.afrag[b][l], byte, 16);
        }
    for (int b = 0; b < 2; b++)
        for (int h = 0; h < 2; h++)
            for (int reg = 0; reg < 16; reg++) {
                const int k = 32 * b + (reg & 3) + 8 * (reg >> 2) + 4 * h;
                int s = 0;
                for (int j = 0; j < 32; j++) { const int idx = k - j; if (idx >= 0 && idx <= 32) s += t[idx]; }
                mc.cinit[b][h][reg] = 128 * s;
            }
    const u32 blocks = 256 * 8, threads = 256, iters = 400;
    const size_t n = (size_t)blocks * threads;
    std::vector<u32> hin(n * 8), hout(n * 16);
    u64 x = 0x243F6A8885A308D3ull;
    for (auto &w : hin) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; w = (u32)(x >> 16); }
    MfmaConst *dmc; u32 *dc, *din, *dout;
    hipMalloc(&dmc, sizeof(mc)); hipMalloc(&dc, 32); hipMalloc(&din, n * 32); hipMalloc(&dout, n * 64);
    hipMemcpy(dmc, &mc, sizeof(mc), hipMemcpyHostToDevice); hipMemcpy(dc, c, 32, hipMemcpyHostToDevice); hipMemcpy(din, hin.data(), n * 32, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; mode++) {
        // exactness first: one product per element against the host
        if (mode == 0) hipLaunchKernelGGL(k_chain<0>, dim3(blocks), dim3(threads), 0, 0, dmc, dc, din, dout, 1);
        else hipLaunchKernelGGL(k_chain<1>, dim3(blocks), dim3(threads), 0, 0, dmc, dc, din, dout, 1);
        hipMemcpy(hout.data(), dout, n * 64, hipMemcpyDeviceToHost);
        size_t bad = 0;
        for (size_t e = 0; e < 4096; e++) { u32 T[16]; host_mul(c, &hin[e * 8], T); if (std::memcmp(T, &hout[e * 16], 64)) bad++; }
        float ms = 0;
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k_chain<0>, dim3(blocks), dim3(threads), 0, 0, dmc, dc, din, dout, iters);
            else hipLaunchKernelGGL(k_chain<1>, dim3(blocks), dim3(threads), 0, 0, dmc, dc, din, dout, iters);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        }
        printf("%-46s %8.3f ms  %.3e constant 256x256->512-bit products/s   mismatches vs host in the first 4096 elements: %zu\n",
               mode == 0 ? "matrix cores (v_mfma_i32_32x32x32_i8)" : "vector ALU (v_mad_u64_u32, constant in SGPRs)", ms, (double)n * iters / (ms * 1e-3), bad);
    }
    printf("(a full Montgomery product is one variable x variable product plus two such constant products, one of them low half only;\n"
           " k_bench_modmul sustains ~1.43e11 of those per second on the vector ALU alone)\n");
    return 0;
}
