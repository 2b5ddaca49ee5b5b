// G1 level-1 bucket accumulation with AFFINE additions and batched inversions ("batch-affine"), nine 29-bit limbs (field29.cuh).
//
// Why.  The XYZZ mixed addition of k_msm_accum_affine29 is 8 products + 2 squares (1467 v_mad_u64_u32) and the kernel runs at >= 90 % of
// what that instruction mix allows: the only way down is fewer multiplications.  An affine addition
//     lambda = (y2 - y1) / (x2 - x1),   x3 = lambda^2 - x1 - x2,   y3 = lambda (x1 - x3) - y1
// is 2 products + 1 square once 1 / (x2 - x1) is known, and Montgomery's trick shares one inversion between many INDEPENDENT additions
// at 3 products each: 5 products + 1 square per addition (936 v_mad_u64_u32) plus the shared part.  gnark-crypto's CPU MultiExp does its
// bucket additions this way too (batch-affine, ecc/bn254 multiexp_affine.go -- not in /root/reference; the result is the same group
// element either way, so the bytes after the final conversion are the same).
//
// Independence.  An item (<= 16 sorted entries of one bucket, msm_core.cuh) is summed as a binary tree instead of a chain: round r adds
// the nodes (r-1, 2j) and (r-1, 2j+1) of every item, j < 16 >> r -- all additions of a round are independent.  Node (r, j) lives at
// nodes[item * 8 + (j << (r-1))], i.e. ON TOP of its left operand, so a node whose right operand does not exist (ragged tail) is already
// in place, and a node that spans one entry is that entry itself (read from the sorted list / the table).  After `rounds` rounds
// k_ba_finish adds what is left of the item (16 >> rounds nodes, XYZZ mixed additions) and writes exactly what the chain kernel writes.
//
// One round = three launches over "slots" (item, j), 64 * K consecutive slots per wave ("chunk"):
//   k_ba_fwd   lane: for its K slots d = x2 - x1, running product of the d's, the product BEFORE each slot parked in prefix[slot]
//              (32 B); the lane's total to totals[chunk][lane].  A slot with an operand at infinity or d = 0 (doubling /
//              cancellation) takes no part: its prefix is the marker 0 and k_ba_bwd adds it in the standard arithmetic.
//   k_ba_inv   thread = chunk: Montgomery's trick over the chunk's 64 lane totals, one Fermat inversion (~330 products) per chunk.
//   k_ba_bwd   lane: backwards over its K slots: 1/d = inv * prefix, inv *= d, then the addition; the sum written over the left operand.
// Shared cost per addition: (63 * 3 + 330) / (64 K) products = 0.25 at K = 32.
//
// Representation.  Coordinates of nodes are stored as 8 x u32 like the table points (value * 2^261 mod p) but NOT canonical: any
// representative below 2.01 p ("almost < 2p", f29_below_2p).  Bounds (V in multiples of p):
//   operands x, y < 2.01      d = x2 + 4p - x1 < 6.02      dy = y2 + 4p - y1 < 6.02      run, inv: products, < 1.1
//   lambda = dy / d < 1.05    lambda^2 < 1.01    x3 = lambda^2 + 8p - (x1 + x2) < 9.01 -> -4p if >= 4p: < 5.01 -> below_2p
//   D = x1 + 4p - x3 < 6.02   lambda D < 1.05    y3 = lambda D + 4p - y1 < 5.05 -> below_2p
#pragma once
#include "curve29.cuh"

#define BA_LOG_L 4u   // items of <= 16 entries

typedef FpParams BAP;
MI_HD F29 ba_one() { return f29_const<BAP>(P29<BAP>::one); }
// x = a product's result (normalised, < 2p): 0 mod p  <=>  x in {0, p}
MI_HD bool ba_is_zero_mod_p(const F29 &x) {
    if (x.l[0] != 0 && x.l[0] != P29<BAP>::p[0]) return false;
    u32 z = 0, e = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) { z |= x.l[i]; e |= x.l[i] ^ P29<BAP>::p[i]; }
    return !z || !e;
}
// an "almost < 2p" representative of a weak value < 6.1 p
MI_HD F29 ba_below_2p(const F29 &x) { return f29_wnorm(f29_condsub(f29_wnorm(f29_condsub(x, P29<BAP>::p4)), P29<BAP>::p2)); }
// x^(p-2) in the R' arithmetic (= the R' form of 1 / x); x weak, V < 8.  0 -> 0.
MI_HD F29 ba_inv(const F29 &x) {
    u32 e[8];
#pragma unroll
    for (int i = 0; i < 8; i++) e[i] = BAP::p[i];
    e[0] -= 2;
    F29 r = x;   // bit 253 of p - 2 is set, bits 254 and 255 are not
    for (int bit = 252; bit >= 0; bit--) {
        r = f29_sqr<BAP>(r);
        if ((e[bit >> 5] >> (bit & 31)) & 1u) r = f29_mul<BAP>(r, x);
    }
    return r;
}
MI_HD void ba_pack(const F29 &x, u32 *w) { f29_pack(f29_norm(x), w); }

// Operand `c` of a round-R addition of an item with entries [b, b + len): node (R-1, c).  NW = 8 loads x only, 16 loads x | y with the
// entry's sign applied to y.  Returns "some word is non-zero" (all zero = the point at infinity for NW = 16).
template <int R, int NW>
static __device__ __forceinline__ u32 ba_fetch(const G1Aff *pts, const u32 *sorted, const uint4 *nodes, u32 item, u32 b, u32 len, u32 c, u32 (&w)[NW]) {
    const u32 start = c << (R - 1);
    const uint4 *src;
    bool neg = false;
    if (R == 1 || start + 1 == len) {   // one entry: the table point itself
        const u32 v = sorted[b + start];
        neg = (v >> 31) != 0;
        src = reinterpret_cast<const uint4 *>(pts + (v & 0x7fffffffu));
    } else {
        src = nodes + ((size_t)item * 8 + (R > 1 ? (c << (R > 1 ? R - 2 : 0)) : 0u)) * 4;
    }
    u32 any = 0;
#pragma unroll
    for (int q = 0; q < NW / 4; q++) {
        const uint4 t = src[q];
        w[4 * q] = t.x; w[4 * q + 1] = t.y; w[4 * q + 2] = t.z; w[4 * q + 3] = t.w;
        any |= t.x | t.y | t.z | t.w;
    }
    if (NW == 16 && neg && any) {   // p - y on the canonical words of a table point
        Fp yy, ny;
#pragma unroll
        for (int i = 0; i < 8; i++) yy.l[i] = w[8 + i];
        fe_sub_raw(ny, Fp::modulus(), yy);
#pragma unroll
        for (int i = 0; i < 8; i++) w[8 + i] = ny.l[i];
    }
    return any;
}

struct BaSlot {
    u32 item, j, b, len;
    bool valid;
};
template <int R>
static __device__ __forceinline__ BaSlot ba_slot(const uint4 *tab, u32 slot, u32 nslots) {
    constexpr u32 SH = BA_LOG_L - R;
    BaSlot s;
    s.valid = false;
    if (slot >= nslots) return s;
    s.item = slot >> SH; s.j = slot & ((1u << SH) - 1u);
    const uint4 rec = tab[s.item];
    s.b = rec.y; s.len = rec.z - rec.y;
    s.valid = ((2 * s.j + 1) << (R - 1)) < s.len;
    return s;
}

template <int R>
__global__ void __launch_bounds__(64) k_ba_fwd(const G1Aff *pts, const u32 *sorted, const uint4 *tab, const u32 *item_start, u32 nkeys, const uint4 *nodes,
                                               uint4 *prefix, uint4 *totals, u32 K) {
    const u32 nslots = item_start[nkeys] << (BA_LOG_L - R), per = 64 * K, nchunks = (nslots + per - 1) / per;
    for (u32 chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        F29 run = ba_one();
        for (u32 i = 0; i < K; i++) {
            const u32 slot = chunk * per + i * 64 + threadIdx.x;
            const BaSlot s = ba_slot<R>(tab, slot, nslots);
            if (!s.valid) continue;
            u32 xa[8], xb[8];
            const u32 anya = ba_fetch<R, 8>(pts, sorted, nodes, s.item, s.b, s.len, 2 * s.j, xa);
            const u32 anyb = ba_fetch<R, 8>(pts, sorted, nodes, s.item, s.b, s.len, 2 * s.j + 1, xb);
            const F29 d = f29_wnorm(f29_sub<BAP>(f29_unpack(xb), f29_unpack(xa), P29<BAP>::c4));
            const F29 nr = f29_mul<BAP>(run, d);
            const bool special = !anya || !anyb || ba_is_zero_mod_p(nr);
            u32 w[8];
            ba_pack(run, w);
            prefix[(size_t)slot * 2] = special ? make_uint4(0, 0, 0, 0) : make_uint4(w[0], w[1], w[2], w[3]);
            prefix[(size_t)slot * 2 + 1] = special ? make_uint4(0, 0, 0, 0) : make_uint4(w[4], w[5], w[6], w[7]);
            if (!special) run = nr;
        }
        u32 w[8];
        ba_pack(run, w);
        totals[((size_t)chunk * 64 + threadIdx.x) * 2] = make_uint4(w[0], w[1], w[2], w[3]);
        totals[((size_t)chunk * 64 + threadIdx.x) * 2 + 1] = make_uint4(w[4], w[5], w[6], w[7]);
    }
}

// thread = chunk: invs[chunk][l] = 1 / totals[chunk][l]
__global__ void __launch_bounds__(64) k_ba_inv(const u32 *item_start, u32 nkeys, u32 sh, u32 K, const u32 *totals, u32 *invs) {
    const u32 nslots = item_start[nkeys] << sh, per = 64 * K, nchunks = (nslots + per - 1) / per;
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nchunks) return;
    const u32 *tin = totals + (size_t)t * 64 * 8;
    u32 *out = invs + (size_t)t * 64 * 8;
    F29 run = ba_one();
    for (u32 l = 0; l < 64; l++) {
        u32 w[8];
        ba_pack(run, w);
#pragma unroll
        for (int i = 0; i < 8; i++) out[l * 8 + i] = w[i];
        run = f29_mul<BAP>(run, f29_unpack(tin + l * 8));
    }
    F29 inv = ba_inv(run);
    for (u32 l = 64; l-- > 0;) {
        const F29 pre = f29_unpack(out + l * 8);
        u32 w[8];
        ba_pack(f29_mul<BAP>(inv, pre), w);
#pragma unroll
        for (int i = 0; i < 8; i++) out[l * 8 + i] = w[i];
        inv = f29_mul<BAP>(inv, f29_unpack(tin + l * 8));
    }
}

// the addition of a slot that took no part in the batch (an operand at infinity, equal x): standard arithmetic, its own inversion
static __device__ __noinline__ void ba_add_special(const u32 *wa, u32 anya, const u32 *wb, u32 anyb, u32 *out) {
    if (!anya || !anyb) {
        const u32 *src = anya ? wa : wb;   // both at infinity: all zero either way
        for (int i = 0; i < 16; i++) out[i] = src[i];
        return;
    }
    const G1Aff a{f29_to_std<BAP>(f29_unpack(wa)), f29_to_std<BAP>(f29_unpack(wa + 8))}, b{f29_to_std<BAP>(f29_unpack(wb)), f29_to_std<BAP>(f29_unpack(wb + 8))};
    G1X s = G1X::from_affine(a);
    xyzz_madd(s, b, false);
    if (s.is_inf()) {
        for (int i = 0; i < 16; i++) out[i] = 0;
        return;
    }
    const G1Aff r = xyzz_to_affine(s);
    ba_pack(f29_from_std<BAP>(r.x), out);
    ba_pack(f29_from_std<BAP>(r.y), out + 8);
}

template <int R>
__global__ void __launch_bounds__(64) k_ba_bwd(const G1Aff *pts, const u32 *sorted, const uint4 *tab, const u32 *item_start, u32 nkeys, uint4 *nodes,
                                               const uint4 *prefix, const uint4 *invs, u32 K) {
    const u32 nslots = item_start[nkeys] << (BA_LOG_L - R), per = 64 * K, nchunks = (nslots + per - 1) / per;
    for (u32 chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        F29 inv;
        {
            const uint4 a = invs[((size_t)chunk * 64 + threadIdx.x) * 2], b = invs[((size_t)chunk * 64 + threadIdx.x) * 2 + 1];
            const u32 w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
            inv = f29_unpack(w);
        }
        for (u32 i = K; i-- > 0;) {
            const u32 slot = chunk * per + i * 64 + threadIdx.x;
            const BaSlot s = ba_slot<R>(tab, slot, nslots);
            if (!s.valid) continue;
            u32 wa[16], wb[16], wo[16];
            const u32 anya = ba_fetch<R, 16>(pts, sorted, nodes, s.item, s.b, s.len, 2 * s.j, wa);
            const u32 anyb = ba_fetch<R, 16>(pts, sorted, nodes, s.item, s.b, s.len, 2 * s.j + 1, wb);
            const uint4 p0 = prefix[(size_t)slot * 2], p1 = prefix[(size_t)slot * 2 + 1];
            const u32 pw[8] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
            u32 pany = 0;
#pragma unroll
            for (int q = 0; q < 8; q++) pany |= pw[q];
            if (!pany) {   // (copies: the arrays handed to the out-of-line function live in scratch memory, wa / wb / wo stay in registers)
                u32 ta[16], tb[16], to[16];
#pragma unroll
                for (int q = 0; q < 16; q++) { ta[q] = wa[q]; tb[q] = wb[q]; }
                ba_add_special(ta, anya, tb, anyb, to);
#pragma unroll
                for (int q = 0; q < 16; q++) wo[q] = to[q];
            } else {
                const F29 x1 = f29_unpack(wa), y1 = f29_unpack(wa + 8), x2 = f29_unpack(wb), y2 = f29_unpack(wb + 8);
                const F29 d = f29_wnorm(f29_sub<BAP>(x2, x1, P29<BAP>::c4));
                const F29 dy = f29_wnorm(f29_sub<BAP>(y2, y1, P29<BAP>::c4));
                const F29 inv_d = f29_mul<BAP>(inv, f29_unpack(pw));
                inv = f29_mul<BAP>(inv, d);
                const F29 lam = f29_mul<BAP>(dy, inv_d);
                const F29 ll = f29_sqr<BAP>(lam);
                const F29 sx = f29_wnorm(f29_add(x1, x2));
                F29 x3 = f29_wnorm(f29_sub<BAP>(ll, sx, P29<BAP>::c8));
                x3 = ba_below_2p(f29_wnorm(f29_condsub(x3, P29<BAP>::p4)));
                const F29 D = f29_wnorm(f29_sub<BAP>(x1, x3, P29<BAP>::c4));
                const F29 m = f29_mul<BAP>(lam, D);
                const F29 y3 = ba_below_2p(f29_wnorm(f29_sub<BAP>(m, y1, P29<BAP>::c4)));
                ba_pack(x3, wo);
                ba_pack(y3, wo + 8);
            }
            uint4 *dst = nodes + ((size_t)s.item * 8 + (s.j << (R - 1))) * 4;
            dst[0] = make_uint4(wo[0], wo[1], wo[2], wo[3]); dst[1] = make_uint4(wo[4], wo[5], wo[6], wo[7]);
            dst[2] = make_uint4(wo[8], wo[9], wo[10], wo[11]); dst[3] = make_uint4(wo[12], wo[13], wo[14], wo[15]);
        }
    }
}

// what is left of every item after RD rounds (ceil(len / 2^RD) nodes) -> the item's sum, written like k_msm_accum_affine29 writes it
template <int RD>
__global__ void __launch_bounds__(64) k_ba_finish(const G1Aff *pts, const u32 *sorted, const uint4 *tab, const u32 *item_start, u32 nkeys, const uint4 *nodes,
                                                  G1X *bucket, G1X *partial_out, u32 rp_partials) {
    const u32 total = item_start[nkeys], stride = gridDim.x * blockDim.x;
    for (u32 item = blockIdx.x * blockDim.x + threadIdx.x; item < total; item += stride) {
        const uint4 rec = tab[item];
        const u32 key = rec.x, b = rec.y, len = rec.z - rec.y, nn = (len + (1u << RD) - 1) >> RD;
        G1X29 acc = g1x29_inf();
        for (u32 k = 0; k < nn; k++) {
            u32 w[16];
            (void)ba_fetch<RD + 1, 16>(pts, sorted, nodes, item, b, len, k, w);
            g1x29_madd(acc, w, false);
        }
        if (rec.w) bucket[key] = g1x29_to_std(acc);
        else if (rp_partials) g1x29_store_rp(acc, reinterpret_cast<u32 *>(partial_out + item));
        else partial_out[item] = g1x29_to_std(acc);
    }
}
