// BN254 Fr / Fp on gfx950: 8 x 32-bit little-endian limbs (same bytes as gnark-crypto's 4 x u64
// fr.Element / fp.Element, typeConverters.go:30-39 limb order), Montgomery form, R = 2^256.
// Replaces gnark-crypto ecc/bn254/{fr,fp} element arithmetic on the path reached from
// /root/reference/mt.go:496.
//
// Multiplication is the "no-carry" CIOS variant (valid because the top limb of both moduli is
// < 2^31): one fused row of x*y[i] and m*p per outer step, written so that hipcc lowers each
// 32x32+64 step to v_mad_u64_u32.  Everything is MI_HD so tests can run the identical code on
// the host (tests/emu), the product only ever runs it on the device.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
#define MI_HD __host__ __device__ __forceinline__
#define MI_D __device__ __forceinline__
#elif defined(__HIPCC__)   // host pass of a .hip file: same functions, ordinary inlining (forced inlining of the whole
#define MI_HD __host__ __device__ inline   // curve arithmetic into the host-side assembly code took minutes to compile)
#define MI_D __device__ inline
#else
#define MI_HD inline
#define MI_D inline
#endif

typedef uint32_t u32;
typedef uint64_t u64;

struct FrParams {
    static constexpr u32 p[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    static constexpr u32 r2[8] = {0xae216da7u, 0x1bb8e645u, 0xe35c59e3u, 0x53fe3ab1u, 0x53bb8085u, 0x8c49833du, 0x7f4e44a5u, 0x0216d0b1u};
    static constexpr u32 one[8] = {0x4ffffffbu, 0xac96341cu, 0x9f60cd29u, 0x36fc7695u, 0x7879462eu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
    static constexpr u32 inv = 0xefffffffu;  // -p^-1 mod 2^32
    static constexpr u32 p2[8] = {0xe0000002u, 0x87c3eb27u, 0xf372e122u, 0x5067d090u, 0x0302b0bau, 0x70a08b6du, 0xc2634053u, 0x60c89ce5u};   // 2p (4p < 2^256 < 6p)
};
struct FpParams {
    static constexpr u32 p[8] = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    static constexpr u32 r2[8] = {0x538afa89u, 0xf32cfc5bu, 0xd44501fbu, 0xb5e71911u, 0x0a417ff6u, 0x47ab1effu, 0xcab8351fu, 0x06d89f71u};
    static constexpr u32 one[8] = {0xc58f0d9du, 0xd35d438du, 0xf5c70b3du, 0x0a78eb28u, 0x7879462cu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
    static constexpr u32 inv = 0xe4866389u;
    static constexpr u32 p2[8] = {0xb0f9fa8eu, 0x7841182du, 0xd0e3951au, 0x2f02d522u, 0x0302b0bbu, 0x70a08b6du, 0xc2634053u, 0x60c89ce5u};   // 2p
};

template <class P>
struct Fe {
    u32 l[8];

    static MI_HD Fe zero() { Fe z; 
#pragma unroll
        for (int i = 0; i < 8; i++) z.l[i] = 0; 
        return z; }
    static MI_HD Fe one() { Fe z; 
#pragma unroll
        for (int i = 0; i < 8; i++) z.l[i] = P::one[i]; 
        return z; }
    static MI_HD Fe r2() { Fe z; 
#pragma unroll
        for (int i = 0; i < 8; i++) z.l[i] = P::r2[i]; 
        return z; }
    static MI_HD Fe modulus() { Fe z; 
#pragma unroll
        for (int i = 0; i < 8; i++) z.l[i] = P::p[i]; 
        return z; }

    MI_HD bool is_zero() const { u32 o = 0; 
#pragma unroll
        for (int i = 0; i < 8; i++) o |= l[i]; 
        return o == 0; }
    MI_HD bool operator==(const Fe &b) const { u32 o = 0; 
#pragma unroll
        for (int i = 0; i < 8; i++) o |= l[i] ^ b.l[i]; 
        return o == 0; }
    MI_HD bool operator!=(const Fe &b) const { return !(*this == b); }
};

// z = x - y, returns borrow
template <class P>
MI_HD u32 fe_sub_raw(Fe<P> &z, const Fe<P> &x, const Fe<P> &y) {
    u64 b = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        u64 d = (u64)x.l[i] - y.l[i] - b;
        z.l[i] = (u32)d;
        b = (d >> 32) & 1;
    }
    return (u32)b;
}
template <class P>
MI_HD u32 fe_add_raw(Fe<P> &z, const Fe<P> &x, const Fe<P> &y) {
    u64 c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (u64)x.l[i] + y.l[i];
        z.l[i] = (u32)c;
        c >>= 32;
    }
    return (u32)c;
}
// Device carry chains are written out (v_add_co/v_addc_co, v_sub_co/v_subb_co + v_cndmask): hipcc lowers the
// portable u64 formulation below to ~80-90 VALU instructions per modular add/sub (64-bit adds, sign-extension
// shifts, pair moves) against 24-25 here.  The host build (tests) keeps the portable code.
#if defined(__HIP_DEVICE_COMPILE__)
#define MI_L8(v) "v"(v.l[0]), "v"(v.l[1]), "v"(v.l[2]), "v"(v.l[3]), "v"(v.l[4]), "v"(v.l[5]), "v"(v.l[6]), "v"(v.l[7])
#define MI_O8(v) "=&v"(v.l[0]), "=&v"(v.l[1]), "=&v"(v.l[2]), "=&v"(v.l[3]), "=&v"(v.l[4]), "=&v"(v.l[5]), "=&v"(v.l[6]), "=&v"(v.l[7])
// the modulus limbs sit in VGPRs here: a VOP2 carry-in (vcc) already uses the one constant-bus read gfx9 allows
#define MI_P8 "v"(P::p[0]), "v"(P::p[1]), "v"(P::p[2]), "v"(P::p[3]), "v"(P::p[4]), "v"(P::p[5]), "v"(P::p[6]), "v"(P::p[7])
// if x >= p: x -= p   (x < 2p)
template <class P>
MI_HD Fe<P> fe_reduce_once(const Fe<P> &x) {
    Fe<P> z;   // operands: %0-7 z, %8-15 x, %16-23 p
    asm("v_subrev_co_u32_e32 %0, vcc, %16, %8\n\t"
        "v_subbrev_co_u32_e32 %1, vcc, %17, %9, vcc\n\t"
        "v_subbrev_co_u32_e32 %2, vcc, %18, %10, vcc\n\t"
        "v_subbrev_co_u32_e32 %3, vcc, %19, %11, vcc\n\t"
        "v_subbrev_co_u32_e32 %4, vcc, %20, %12, vcc\n\t"
        "v_subbrev_co_u32_e32 %5, vcc, %21, %13, vcc\n\t"
        "v_subbrev_co_u32_e32 %6, vcc, %22, %14, vcc\n\t"
        "v_subbrev_co_u32_e32 %7, vcc, %23, %15, vcc\n\t"
        "v_cndmask_b32_e32 %0, %0, %8, vcc\n\t"
        "v_cndmask_b32_e32 %1, %1, %9, vcc\n\t"
        "v_cndmask_b32_e32 %2, %2, %10, vcc\n\t"
        "v_cndmask_b32_e32 %3, %3, %11, vcc\n\t"
        "v_cndmask_b32_e32 %4, %4, %12, vcc\n\t"
        "v_cndmask_b32_e32 %5, %5, %13, vcc\n\t"
        "v_cndmask_b32_e32 %6, %6, %14, vcc\n\t"
        "v_cndmask_b32_e32 %7, %7, %15, vcc"
        : MI_O8(z) : MI_L8(x), MI_P8 : "vcc");
    return z;
}
template <class P>
MI_HD Fe<P> operator+(const Fe<P> &x, const Fe<P> &y) {
    Fe<P> z, t;   // %0-7 z, %8-15 t, %16-23 x, %24-31 y, %32-39 p.   t = x + y (no carry out: p < 2^254); z = t - p; pick
    asm("v_add_co_u32_e32 %8, vcc, %16, %24\n\t"
        "v_addc_co_u32_e32 %9, vcc, %17, %25, vcc\n\t"
        "v_addc_co_u32_e32 %10, vcc, %18, %26, vcc\n\t"
        "v_addc_co_u32_e32 %11, vcc, %19, %27, vcc\n\t"
        "v_addc_co_u32_e32 %12, vcc, %20, %28, vcc\n\t"
        "v_addc_co_u32_e32 %13, vcc, %21, %29, vcc\n\t"
        "v_addc_co_u32_e32 %14, vcc, %22, %30, vcc\n\t"
        "v_addc_co_u32_e32 %15, vcc, %23, %31, vcc\n\t"
        "v_subrev_co_u32_e32 %0, vcc, %32, %8\n\t"
        "v_subbrev_co_u32_e32 %1, vcc, %33, %9, vcc\n\t"
        "v_subbrev_co_u32_e32 %2, vcc, %34, %10, vcc\n\t"
        "v_subbrev_co_u32_e32 %3, vcc, %35, %11, vcc\n\t"
        "v_subbrev_co_u32_e32 %4, vcc, %36, %12, vcc\n\t"
        "v_subbrev_co_u32_e32 %5, vcc, %37, %13, vcc\n\t"
        "v_subbrev_co_u32_e32 %6, vcc, %38, %14, vcc\n\t"
        "v_subbrev_co_u32_e32 %7, vcc, %39, %15, vcc\n\t"
        "v_cndmask_b32_e32 %0, %0, %8, vcc\n\t"
        "v_cndmask_b32_e32 %1, %1, %9, vcc\n\t"
        "v_cndmask_b32_e32 %2, %2, %10, vcc\n\t"
        "v_cndmask_b32_e32 %3, %3, %11, vcc\n\t"
        "v_cndmask_b32_e32 %4, %4, %12, vcc\n\t"
        "v_cndmask_b32_e32 %5, %5, %13, vcc\n\t"
        "v_cndmask_b32_e32 %6, %6, %14, vcc\n\t"
        "v_cndmask_b32_e32 %7, %7, %15, vcc"
        : MI_O8(z), MI_O8(t) : MI_L8(x), MI_L8(y), MI_P8 : "vcc");
    return z;
}
template <class P>
MI_HD Fe<P> operator-(const Fe<P> &x, const Fe<P> &y) {
    Fe<P> z, t;   // %0-7 z, %8-15 t, %16 mask, %17-24 x, %25-32 y, %33-40 p.   z = x - y; mask = borrow ? ~0 : 0; z += p & mask
    u32 mask;
    asm("v_sub_co_u32_e32 %0, vcc, %17, %25\n\t"
        "v_subb_co_u32_e32 %1, vcc, %18, %26, vcc\n\t"
        "v_subb_co_u32_e32 %2, vcc, %19, %27, vcc\n\t"
        "v_subb_co_u32_e32 %3, vcc, %20, %28, vcc\n\t"
        "v_subb_co_u32_e32 %4, vcc, %21, %29, vcc\n\t"
        "v_subb_co_u32_e32 %5, vcc, %22, %30, vcc\n\t"
        "v_subb_co_u32_e32 %6, vcc, %23, %31, vcc\n\t"
        "v_subb_co_u32_e32 %7, vcc, %24, %32, vcc\n\t"
        "v_cndmask_b32_e64 %16, 0, -1, vcc\n\t"
        "v_and_b32_e32 %8, %33, %16\n\t"
        "v_and_b32_e32 %9, %34, %16\n\t"
        "v_and_b32_e32 %10, %35, %16\n\t"
        "v_and_b32_e32 %11, %36, %16\n\t"
        "v_and_b32_e32 %12, %37, %16\n\t"
        "v_and_b32_e32 %13, %38, %16\n\t"
        "v_and_b32_e32 %14, %39, %16\n\t"
        "v_and_b32_e32 %15, %40, %16\n\t"
        "v_add_co_u32_e32 %0, vcc, %0, %8\n\t"
        "v_addc_co_u32_e32 %1, vcc, %1, %9, vcc\n\t"
        "v_addc_co_u32_e32 %2, vcc, %2, %10, vcc\n\t"
        "v_addc_co_u32_e32 %3, vcc, %3, %11, vcc\n\t"
        "v_addc_co_u32_e32 %4, vcc, %4, %12, vcc\n\t"
        "v_addc_co_u32_e32 %5, vcc, %5, %13, vcc\n\t"
        "v_addc_co_u32_e32 %6, vcc, %6, %14, vcc\n\t"
        "v_addc_co_u32_e32 %7, vcc, %7, %15, vcc"
        : MI_O8(z), MI_O8(t), "=&v"(mask) : MI_L8(x), MI_L8(y), MI_P8 : "vcc");
    return z;
}
#undef MI_L8
#undef MI_O8
#undef MI_P8
#else
// if x >= p: x -= p   (x < 2p)
template <class P>
MI_HD Fe<P> fe_reduce_once(const Fe<P> &x) {
    Fe<P> d;
    u32 borrow = fe_sub_raw(d, x, Fe<P>::modulus());
    Fe<P> z;
#pragma unroll
    for (int i = 0; i < 8; i++) z.l[i] = borrow ? x.l[i] : d.l[i];
    return z;
}
template <class P>
MI_HD Fe<P> operator+(const Fe<P> &x, const Fe<P> &y) {
    Fe<P> s;
    fe_add_raw(s, x, y);  // p < 2^254: no carry out
    return fe_reduce_once(s);
}
template <class P>
MI_HD Fe<P> operator-(const Fe<P> &x, const Fe<P> &y) {
    Fe<P> d, e;
    u32 borrow = fe_sub_raw(d, x, y);
    fe_add_raw(e, d, Fe<P>::modulus());
    Fe<P> z;
#pragma unroll
    for (int i = 0; i < 8; i++) z.l[i] = borrow ? e.l[i] : d.l[i];
    return z;
}
#endif
// ---------------------------------------------------------------- lazy arithmetic (the NTT butterflies, ntt_tile.cuh)
// Values are kept as 256-bit representatives in [0, 2p) or [0, 4p) instead of [0, p): 4p < 2^256 for both moduli, so plain 8-limb
// additions cannot carry out, and a Montgomery product of x < 4p by a table constant w < p comes out below x w / R + p < 1.76 p < 2p
// WITHOUT its final conditional subtraction (fe_mul_lazy).  Harvey's butterflies (ntt_bfly_dif / ntt_bfly_dit) then need one
// conditional subtraction of 2p where the classical ones need three conditional corrections by p.  Same residues; the last pass of a
// transform brings every element back to [0, p) (fe_canon), so results are bit-identical.
#if defined(__HIP_DEVICE_COMPILE__)
#define MI_L8(v) "v"(v.l[0]), "v"(v.l[1]), "v"(v.l[2]), "v"(v.l[3]), "v"(v.l[4]), "v"(v.l[5]), "v"(v.l[6]), "v"(v.l[7])
#define MI_O8(v) "=&v"(v.l[0]), "=&v"(v.l[1]), "=&v"(v.l[2]), "=&v"(v.l[3]), "=&v"(v.l[4]), "=&v"(v.l[5]), "=&v"(v.l[6]), "=&v"(v.l[7])
#define MI_2P8 "v"(P::p2[0]), "v"(P::p2[1]), "v"(P::p2[2]), "v"(P::p2[3]), "v"(P::p2[4]), "v"(P::p2[5]), "v"(P::p2[6]), "v"(P::p2[7])
// x + y, no reduction (the caller knows the sum stays below 2^256)
template <class P>
MI_HD Fe<P> fe_add_nored(const Fe<P> &x, const Fe<P> &y) {
    Fe<P> z;   // %0-7 z, %8-15 x, %16-23 y
    asm("v_add_co_u32_e32 %0, vcc, %8, %16\n\t"
        "v_addc_co_u32_e32 %1, vcc, %9, %17, vcc\n\t"
        "v_addc_co_u32_e32 %2, vcc, %10, %18, vcc\n\t"
        "v_addc_co_u32_e32 %3, vcc, %11, %19, vcc\n\t"
        "v_addc_co_u32_e32 %4, vcc, %12, %20, vcc\n\t"
        "v_addc_co_u32_e32 %5, vcc, %13, %21, vcc\n\t"
        "v_addc_co_u32_e32 %6, vcc, %14, %22, vcc\n\t"
        "v_addc_co_u32_e32 %7, vcc, %15, %23, vcc"
        : MI_O8(z) : MI_L8(x), MI_L8(y) : "vcc");
    return z;
}
// x - y + 2p, no reduction: in (0, 4p) for x, y in [0, 2p)
template <class P>
MI_HD Fe<P> fe_sub_plus2p(const Fe<P> &x, const Fe<P> &y) {
    Fe<P> z;   // %0-7 z, %8-15 x, %16-23 y, %24-31 2p
    asm("v_sub_co_u32_e32 %0, vcc, %8, %16\n\t"
        "v_subb_co_u32_e32 %1, vcc, %9, %17, vcc\n\t"
        "v_subb_co_u32_e32 %2, vcc, %10, %18, vcc\n\t"
        "v_subb_co_u32_e32 %3, vcc, %11, %19, vcc\n\t"
        "v_subb_co_u32_e32 %4, vcc, %12, %20, vcc\n\t"
        "v_subb_co_u32_e32 %5, vcc, %13, %21, vcc\n\t"
        "v_subb_co_u32_e32 %6, vcc, %14, %22, vcc\n\t"
        "v_subb_co_u32_e32 %7, vcc, %15, %23, vcc\n\t"
        "v_add_co_u32_e32 %0, vcc, %0, %24\n\t"
        "v_addc_co_u32_e32 %1, vcc, %1, %25, vcc\n\t"
        "v_addc_co_u32_e32 %2, vcc, %2, %26, vcc\n\t"
        "v_addc_co_u32_e32 %3, vcc, %3, %27, vcc\n\t"
        "v_addc_co_u32_e32 %4, vcc, %4, %28, vcc\n\t"
        "v_addc_co_u32_e32 %5, vcc, %5, %29, vcc\n\t"
        "v_addc_co_u32_e32 %6, vcc, %6, %30, vcc\n\t"
        "v_addc_co_u32_e32 %7, vcc, %7, %31, vcc"
        : MI_O8(z) : MI_L8(x), MI_L8(y), MI_2P8 : "vcc");
    return z;
}
// if x >= 2p: x -= 2p   (x < 4p -> [0, 2p))
template <class P>
MI_HD Fe<P> fe_condsub_2p(const Fe<P> &x) {
    Fe<P> z;   // %0-7 z, %8-15 x, %16-23 2p
    asm("v_subrev_co_u32_e32 %0, vcc, %16, %8\n\t"
        "v_subbrev_co_u32_e32 %1, vcc, %17, %9, vcc\n\t"
        "v_subbrev_co_u32_e32 %2, vcc, %18, %10, vcc\n\t"
        "v_subbrev_co_u32_e32 %3, vcc, %19, %11, vcc\n\t"
        "v_subbrev_co_u32_e32 %4, vcc, %20, %12, vcc\n\t"
        "v_subbrev_co_u32_e32 %5, vcc, %21, %13, vcc\n\t"
        "v_subbrev_co_u32_e32 %6, vcc, %22, %14, vcc\n\t"
        "v_subbrev_co_u32_e32 %7, vcc, %23, %15, vcc\n\t"
        "v_cndmask_b32_e32 %0, %0, %8, vcc\n\t"
        "v_cndmask_b32_e32 %1, %1, %9, vcc\n\t"
        "v_cndmask_b32_e32 %2, %2, %10, vcc\n\t"
        "v_cndmask_b32_e32 %3, %3, %11, vcc\n\t"
        "v_cndmask_b32_e32 %4, %4, %12, vcc\n\t"
        "v_cndmask_b32_e32 %5, %5, %13, vcc\n\t"
        "v_cndmask_b32_e32 %6, %6, %14, vcc\n\t"
        "v_cndmask_b32_e32 %7, %7, %15, vcc"
        : MI_O8(z) : MI_L8(x), MI_2P8 : "vcc");
    return z;
}
#undef MI_L8
#undef MI_O8
#undef MI_2P8
#else
template <class P>
MI_HD Fe<P> fe_add_nored(const Fe<P> &x, const Fe<P> &y) {
    Fe<P> z;
    const u32 carry = fe_add_raw(z, x, y);
#if defined(MI_CHECK_NOWRAP)
    if (carry) __builtin_trap();   // the host build of the tests checks every bound the lazy arithmetic relies on
#else
    (void)carry;
#endif
    return z;
}
template <class P>
MI_HD Fe<P> fe_twice_modulus() { Fe<P> z;
#pragma unroll
    for (int i = 0; i < 8; i++) z.l[i] = P::p2[i];
    return z; }
template <class P>
MI_HD Fe<P> fe_sub_plus2p(const Fe<P> &x, const Fe<P> &y) {
    Fe<P> d, z;
    const u32 borrow = fe_sub_raw(d, x, y);
    const u32 carry = fe_add_raw(z, d, fe_twice_modulus<P>());
#if defined(MI_CHECK_NOWRAP)
    if (borrow != carry) __builtin_trap();   // x - y + 2p must land in [0, 2^256): a borrow is always paid back, no borrow never carries
#else
    (void)borrow; (void)carry;
#endif
    return z;
}
template <class P>
MI_HD Fe<P> fe_condsub_2p(const Fe<P> &x) {
    Fe<P> d, z;
    const u32 borrow = fe_sub_raw(d, x, fe_twice_modulus<P>());
#pragma unroll
    for (int i = 0; i < 8; i++) z.l[i] = borrow ? x.l[i] : d.l[i];
    return z;
}
#endif
// any representative below 4p -> the canonical one in [0, p)
template <class P>
MI_HD Fe<P> fe_canon(const Fe<P> &x) { return fe_reduce_once(fe_condsub_2p(x)); }

template <class P>
MI_HD Fe<P> fe_neg(const Fe<P> &x) {
    Fe<P> d;
    fe_sub_raw(d, Fe<P>::modulus(), x);
    bool zr = x.is_zero();
#pragma unroll
    for (int i = 0; i < 8; i++) d.l[i] = zr ? 0u : d.l[i];
    return d;
}
template <class P>
MI_HD Fe<P> fe_dbl(const Fe<P> &x) { return x + x; }

// ---- 96-bit column accumulator primitive: (c : acc) += a * b
// Device: one v_mad_u64_u32 into an even-aligned 64-bit pair plus one v_addc for the carry word; the
// operands never leave their registers (the compiler's own lowering of the row-wise CIOS spent ~350
// v_mov per product re-aligning 64-bit pairs).  Host (tests): the same arithmetic in plain C++.
MI_HD void mac96(u64 &acc, u32 &c, u32 a, u32 b) {
#if defined(__HIP_DEVICE_COMPILE__)
    u64 cy;
    asm("v_mad_u64_u32 %0, %1, %3, %4, %0\n\tv_addc_co_u32_e64 %2, %1, 0, %2, %1"
        : "+v"(acc), "=&s"(cy), "+v"(c)
        : "v"(a), "v"(b));
#else
    u64 p = (u64)a * b;
    acc += p;
    c += acc < p;
#endif
}
// modulus limb as the second factor: kept in an SGPR on the device (one scalar operand is legal)
MI_HD void mac96_k(u64 &acc, u32 &c, u32 a, u32 k) {
#if defined(__HIP_DEVICE_COMPILE__)
    u64 cy;
    asm("v_mad_u64_u32 %0, %1, %3, %4, %0\n\tv_addc_co_u32_e64 %2, %1, 0, %2, %1"
        : "+v"(acc), "=&s"(cy), "+v"(c)
        : "v"(a), "s"(k));
#else
    mac96(acc, c, a, k);
#endif
}
// first product of a column, added without a carry-word update.  Only legal where the sum provably cannot wrap:
// acc < 25 * 2^32 on entry, so the product must stay < 2^63 -- column 0 (acc = 0), m_(k-1)*p_1 in columns 1..7
// (p_1 < 2^31 for both moduli), x_(k-7)*y_7 in columns >= 8 (top limb of any operand < 2p is < 2^31).  x_0*y_k is NOT
// (round 1 used it: wrong products for limbs near 2^32, ~2^-60 per random product).  The host build of the tests
// checks the no-wrap claim on every call (MI_CHECK_NOWRAP).
MI_HD void mac96_first(u64 &acc, u32 a, u32 b) {
#if defined(__HIP_DEVICE_COMPILE__)
    u64 cy;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(acc), "=&s"(cy) : "v"(a), "v"(b));
#else
    u64 p = (u64)a * b;
    acc += p;
#if defined(MI_CHECK_NOWRAP)
    if (acc < p) __builtin_trap();
#endif
#endif
}

// Montgomery product x*y/R mod p: product scanning (column-wise, "FIPS") over 32-bit limbs with a
// 96-bit accumulator.  Column k collects x_i*y_(k-i) and m_i*p_(k-i); m_k makes the column's low word 0.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MI_MONT_PER_MAC)
#include "mont_cols.inc"   // generated: one asm block per column (tools/gen_mont_cols.py)
#define MI_MONT_LO(k)                                   \
    mont_col##k<P>(acc, c, x, y, m);                    \
    m[k] = (u32)acc * P::inv;                           \
    mac96_k(acc, c, m[k], P::p[0]);                     \
    acc = (acc >> 32) | ((u64)c << 32);                 \
    c = 0;
#define MI_MONT_HI(k)                                   \
    mont_col##k<P>(acc, c, x, y, m);                    \
    r.l[k - 8] = (u32)acc;                              \
    acc = (acc >> 32) | ((u64)c << 32);                 \
    c = 0;
template <class P>
MI_HD Fe<P> operator*(const Fe<P> &x, const Fe<P> &y) {
    u64 acc = 0;
    u32 c = 0;
    u32 m[8];
    Fe<P> r;
    MI_MONT_LO(0) MI_MONT_LO(1) MI_MONT_LO(2) MI_MONT_LO(3) MI_MONT_LO(4) MI_MONT_LO(5) MI_MONT_LO(6) MI_MONT_LO(7)
    MI_MONT_HI(8) MI_MONT_HI(9) MI_MONT_HI(10) MI_MONT_HI(11) MI_MONT_HI(12) MI_MONT_HI(13) MI_MONT_HI(14)
    r.l[7] = (u32)acc;   // column 15 is empty for 8-limb operands; p < 2^254 keeps the result < 2p < 2^256
    return fe_reduce_once(r);
}
// the same product without its final conditional subtraction: x < 4p (any limbs), y < p (a table constant: top limb < 2^30, what
// the carry-less first products of columns >= 8 need) -> a representative below x y / R + p < 2p
template <class P>
MI_HD Fe<P> fe_mul_lazy(const Fe<P> &x, const Fe<P> &y) {
    u64 acc = 0;
    u32 c = 0;
    u32 m[8];
    Fe<P> r;
    MI_MONT_LO(0) MI_MONT_LO(1) MI_MONT_LO(2) MI_MONT_LO(3) MI_MONT_LO(4) MI_MONT_LO(5) MI_MONT_LO(6) MI_MONT_LO(7)
    MI_MONT_HI(8) MI_MONT_HI(9) MI_MONT_HI(10) MI_MONT_HI(11) MI_MONT_HI(12) MI_MONT_HI(13) MI_MONT_HI(14)
    r.l[7] = (u32)acc;
    return r;
}
#undef MI_MONT_LO
#undef MI_MONT_HI
#else
template <class P>
MI_HD Fe<P> fe_mul_lazy(const Fe<P> &x, const Fe<P> &y) {   // the product before its final conditional subtraction (see the device form)
    u64 acc = 0;
    u32 c = 0;
    u32 m[8];
    Fe<P> r;
#pragma unroll
    for (int k = 0; k < 8; k++) {   // same product order as the generated device columns (tools/gen_mont_cols.py)
        if (k == 0) mac96_first(acc, x.l[0], y.l[0]);
        else mac96_first(acc, m[k - 1], P::p[1]);
#pragma unroll
        for (int i = k ? 0 : 1; i <= k; i++) mac96(acc, c, x.l[i], y.l[k - i]);
#pragma unroll
        for (int i = 0; i + 1 < k; i++) mac96_k(acc, c, m[i], P::p[k - i]);
        m[k] = (u32)acc * P::inv;
        mac96_k(acc, c, m[k], P::p[0]);
        acc = (acc >> 32) | ((u64)c << 32);
        c = 0;
    }
#pragma unroll
    for (int k = 8; k < 15; k++) {
        mac96_first(acc, x.l[k - 7], y.l[7]);
#pragma unroll
        for (int i = k - 6; i < 8; i++) mac96(acc, c, x.l[i], y.l[k - i]);
#pragma unroll
        for (int i = k - 7; i < 8; i++) mac96_k(acc, c, m[i], P::p[k - i]);
        r.l[k - 8] = (u32)acc;
        acc = (acc >> 32) | ((u64)c << 32);
        c = 0;
    }
    r.l[7] = (u32)acc;
#if defined(MI_CHECK_NOWRAP)
    if (acc >> 32) __builtin_trap();   // the result must fit 256 bits (x y / R + p < 2^256)
#endif
    return r;
}
template <class P>
MI_HD Fe<P> operator*(const Fe<P> &x, const Fe<P> &y) { return fe_reduce_once(fe_mul_lazy(x, y)); }
#endif
// (x*y + u*v) / R mod p with ONE Montgomery reduction ("lazy reduction" of a sum of two products): 128 + 72 mads instead
// of 2 * 136.  Inputs may be <= p (a raw p - a is accepted as the negation of a), the result is canonical.
// Bound: (xy + uv + mp) / R < (2p^2 + Rp) / R < 2p because 2p < R.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MI_MONT_PER_MAC)
#define MI_MONT2_LO(k)                                  \
    mont2_col##k<P>(acc, c, x, y, u, v, m);             \
    m[k] = (u32)acc * P::inv;                           \
    mac96_k(acc, c, m[k], P::p[0]);                     \
    acc = (acc >> 32) | ((u64)c << 32);                 \
    c = 0;
#define MI_MONT2_HI(k)                                  \
    mont2_col##k<P>(acc, c, x, y, u, v, m);             \
    r.l[k - 8] = (u32)acc;                              \
    acc = (acc >> 32) | ((u64)c << 32);                 \
    c = 0;
template <class P>
MI_HD Fe<P> fe_mul2_add(const Fe<P> &x, const Fe<P> &y, const Fe<P> &u, const Fe<P> &v) {
    u64 acc = 0;
    u32 c = 0;
    u32 m[8];
    Fe<P> r;
    MI_MONT2_LO(0) MI_MONT2_LO(1) MI_MONT2_LO(2) MI_MONT2_LO(3) MI_MONT2_LO(4) MI_MONT2_LO(5) MI_MONT2_LO(6) MI_MONT2_LO(7)
    MI_MONT2_HI(8) MI_MONT2_HI(9) MI_MONT2_HI(10) MI_MONT2_HI(11) MI_MONT2_HI(12) MI_MONT2_HI(13) MI_MONT2_HI(14)
    r.l[7] = (u32)acc;
    return fe_reduce_once(r);
}
#undef MI_MONT2_LO
#undef MI_MONT2_HI
#else
template <class P>
MI_HD Fe<P> fe_mul2_add(const Fe<P> &x, const Fe<P> &y, const Fe<P> &u, const Fe<P> &v) {
    u64 acc = 0;
    u32 c = 0;
    u32 m[8];
    Fe<P> r;
    for (int k = 0; k < 15; k++) {   // first (carry-less) product of each column as on the device, see mac96_first
        int lo = k > 7 ? k - 7 : 0, hi = k < 7 ? k : 7;
        if (k == 0) mac96_first(acc, x.l[0], y.l[0]);
        else if (k < 8) mac96_first(acc, m[k - 1], P::p[1]);
        else mac96_first(acc, x.l[k - 7], y.l[7]);
        for (int i = lo; i <= hi; i++) {
            if (!(k == 0 || (k >= 8 && i == lo))) mac96(acc, c, x.l[i], y.l[k - i]);
            mac96(acc, c, u.l[i], v.l[k - i]);
        }
        for (int i = lo; i <= (k < 8 ? k - 2 : 7); i++) mac96(acc, c, m[i], P::p[k - i]);
        if (k < 8) { m[k] = (u32)acc * P::inv; mac96(acc, c, m[k], P::p[0]); } else r.l[k - 8] = (u32)acc;
        acc = (acc >> 32) | ((u64)c << 32);
        c = 0;
    }
    r.l[7] = (u32)acc;
    return fe_reduce_once(r);
}
#endif
// p - x without the zero fix-up (0 -> p): a valid factor for fe_mul2_add / operator*, NOT a canonical value
template <class P>
MI_HD Fe<P> fe_neg_raw(const Fe<P> &x) {
    Fe<P> d;
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_sub_co_u32_e32 %0, vcc, %16, %8\n\t"
        "v_subb_co_u32_e32 %1, vcc, %17, %9, vcc\n\t"
        "v_subb_co_u32_e32 %2, vcc, %18, %10, vcc\n\t"
        "v_subb_co_u32_e32 %3, vcc, %19, %11, vcc\n\t"
        "v_subb_co_u32_e32 %4, vcc, %20, %12, vcc\n\t"
        "v_subb_co_u32_e32 %5, vcc, %21, %13, vcc\n\t"
        "v_subb_co_u32_e32 %6, vcc, %22, %14, vcc\n\t"
        "v_subb_co_u32_e32 %7, vcc, %23, %15, vcc"
        : "=&v"(d.l[0]), "=&v"(d.l[1]), "=&v"(d.l[2]), "=&v"(d.l[3]), "=&v"(d.l[4]), "=&v"(d.l[5]), "=&v"(d.l[6]), "=&v"(d.l[7])
        : "v"(x.l[0]), "v"(x.l[1]), "v"(x.l[2]), "v"(x.l[3]), "v"(x.l[4]), "v"(x.l[5]), "v"(x.l[6]), "v"(x.l[7]),
          "v"(P::p[0]), "v"(P::p[1]), "v"(P::p[2]), "v"(P::p[3]), "v"(P::p[4]), "v"(P::p[5]), "v"(P::p[6]), "v"(P::p[7])
        : "vcc");
#else
    fe_sub_raw(d, Fe<P>::modulus(), x);
#endif
    return d;
}
// a*b - c*d with one reduction
template <class P>
MI_HD Fe<P> fe_mul_sub(const Fe<P> &a, const Fe<P> &b, const Fe<P> &c, const Fe<P> &d) { return fe_mul2_add(a, b, fe_neg_raw(c), d); }
template <class P>
MI_HD Fe<P> fe_sqr(const Fe<P> &x) { return x * x; }

template <class P>
MI_HD Fe<P> fe_to_mont(const Fe<P> &x) { return x * Fe<P>::r2(); }
template <class P>
MI_HD Fe<P> fe_from_mont(const Fe<P> &x) {
    Fe<P> o = Fe<P>::zero();
    o.l[0] = 1;
    return x * o;
}
// x^e, e = 8 x u32 little-endian plain integer
template <class P>
MI_HD Fe<P> fe_pow(const Fe<P> &x, const u32 e[8]) {
    Fe<P> acc = Fe<P>::one();
    for (int i = 255; i >= 0; i--) {
        acc = fe_sqr(acc);
        if ((e[i >> 5] >> (i & 31)) & 1) acc = acc * x;
    }
    return acc;
}
template <class P>
MI_HD Fe<P> fe_inv(const Fe<P> &x) {  // Fermat, 0 -> 0
    u32 e[8];
#pragma unroll
    for (int i = 0; i < 8; i++) e[i] = P::p[i];
    e[0] -= 2;  // low limb of both moduli is >= 2
    return fe_pow(x, e);
}
template <class P>
MI_HD Fe<P> fe_from_u32(u32 v) {
    Fe<P> t = Fe<P>::zero();
    t.l[0] = v;
    return fe_to_mont(t);
}

typedef Fe<FrParams> Fr;
typedef Fe<FpParams> Fp;

// ---------------------------------------------------------------- Fp2 = Fp[u]/(u^2+1)  (fptower.E2)
struct Fp2 {
    Fp a0, a1;
    static MI_HD Fp2 zero() { return Fp2{Fp::zero(), Fp::zero()}; }
    static MI_HD Fp2 one() { return Fp2{Fp::one(), Fp::zero()}; }
    MI_HD bool is_zero() const { return a0.is_zero() && a1.is_zero(); }
    MI_HD bool operator==(const Fp2 &b) const { return a0 == b.a0 && a1 == b.a1; }
    MI_HD bool operator!=(const Fp2 &b) const { return !(*this == b); }
};
MI_HD Fp2 operator+(const Fp2 &x, const Fp2 &y) { return Fp2{x.a0 + y.a0, x.a1 + y.a1}; }
MI_HD Fp2 operator-(const Fp2 &x, const Fp2 &y) { return Fp2{x.a0 - y.a0, x.a1 - y.a1}; }
MI_HD Fp2 fe_neg(const Fp2 &x) { return Fp2{fe_neg(x.a0), fe_neg(x.a1)}; }
MI_HD Fp2 fe_dbl(const Fp2 &x) { return x + x; }
// Fp products inside Fp2 go through ONE out-of-line copy of the multiplier on the device: a G2 mixed
// addition is ~28 Fp products; fully inlined that is > 100 KB of straight-line code per kernel, more than the
// instruction cache holds.  (The G1 path keeps its ~10 products inline.)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MI_FP2_INLINE_MUL)
__device__ __attribute__((noinline)) Fp fp_mul_call(Fp x, Fp y) { return x * y; }
#else
MI_HD Fp fp_mul_call(const Fp &x, const Fp &y) { return x * y; }
#endif
// Karatsuba: 3 base multiplications.  (Schoolbook with lazy reduction -- two fe_mul2_add of 200 mads each -- was measured
// slower here: 3.45 vs 3.96 G mixed additions/s, the four-operand out-of-line call costs more than the five additions it saves.)
MI_HD Fp2 operator*(const Fp2 &x, const Fp2 &y) {
    Fp v0 = fp_mul_call(x.a0, y.a0), v1 = fp_mul_call(x.a1, y.a1);
    Fp s = fp_mul_call(x.a0 + x.a1, y.a0 + y.a1);
    return Fp2{v0 - v1, s - v0 - v1};
}
// (a0+a1)(a0-a1), 2 a0 a1 : 2 base multiplications
MI_HD Fp2 fe_sqr(const Fp2 &x) {
    Fp m = fp_mul_call(x.a0, x.a1);
    return Fp2{fp_mul_call(x.a0 + x.a1, x.a0 - x.a1), m + m};
}
MI_HD Fp2 fe_mul_sub(const Fp2 &a, const Fp2 &b, const Fp2 &c, const Fp2 &d) { return a * b - c * d; }
MI_HD Fp2 fe_inv(const Fp2 &x) {
    Fp n = fe_inv(fe_sqr(x.a0) + fe_sqr(x.a1));
    return Fp2{x.a0 * n, fe_neg(x.a1 * n)};
}
