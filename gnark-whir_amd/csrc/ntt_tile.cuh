// One pass of the multi-pass radix-2 NTT over Fr, expressed as three per-thread phases
// (load / stage / store) over an LDS-resident tile, so that the HIP kernel (ntt.hip) and the
// host emulation used by the CPU tests run the identical index arithmetic.
//
// Replaces gnark-crypto ecc/bn254/fr/fft difFFT / ditFFT (+ the OnCoset scaling loops of
// Domain.FFT / FFTInverse) on the computeH path reached from /root/reference/mt.go:496
// (SURVEY.md 8a rows a3/a4).  Same results, different decomposition:
//
//   N = R_1 * R_2 * ... * R_m.  Pass t views the array as blocks of M = R*S elements
//   (S = stride = product of the later radices for DIF, of the earlier ones for DIT); inside a
//   block, element (row rho, column lo) sits at rho*S + lo.  A tile is all R rows x C columns.
//     DIF pass:  size-R DIF over the rows (natural in, bit-reversed out), then row rho is
//                multiplied by w_M^(lo * bitrev_R(rho)).
//     DIT pass:  row rho is first multiplied by w_M^(lo * bitrev_R(rho)), then size-R DIT over
//                the rows (bit-reversed in, natural out).
//   This is the Cooley-Tukey regrouping of the radix-2 stage twiddles (exact in a field), so the
//   output equals gnark's stage-by-stage result bit for bit.  Coset shifts and 1/N are folded
//   into the first pass's load or the last pass's store.
//
// LDS image: two planes of 16-byte halves (lo / hi 128 bits of each element) so that a wave's
// ds_read_b128 / ds_write_b128 walk consecutive 16-byte slots (conflict-free at unit stride).
#pragma once
#include "field.cuh"

struct alignas(16) U4 { u32 x, y, z, w; };  // 16-byte LDS slot (uint4 without pulling hip headers into host builds); the alignment is what lets
                                            // the compiler use ds_read_b128 / ds_write_b128 (round 1's unaligned struct compiled to ds_read2_b32 pairs)

struct NttTables {
    const Fr *small;    // small[j] = w_4096^j, j < 2048  (in-tile butterfly twiddles for every radix <= 2^12)
    const Fr *tw_lo;    // tw_lo[j] = w_N^j,          j < 2^tw_h
    const Fr *tw_hi;    // tw_hi[j] = w_N^(j * 2^tw_h)
    const Fr *sc_lo;    // scale tables: sc_lo[j] = g^j, sc_hi[j] = const * g^(j * 2^tw_h)  (g = 5 or 1/5)
    const Fr *sc_hi;
    const Fr *tw_64k;   // tw_64k[j] = w_65536^j, j < 65536: blocks of M <= 2^16 read their inter-pass twiddle directly (no product)
    u32 tw_h;
};

struct NttPass {
    u32 log_n;      // transform size
    u32 log_r;      // radix of this pass (rows per tile)
    u32 log_s;      // stride between rows: element (rho, lo) of block hi is at hi*R*S + rho*S + lo
    u32 log_c;      // columns per tile (C <= S) -- for S == 1 the C "columns" are consecutive blocks
    u32 dit;        // 0: DIF butterflies + post twiddle; 1: pre twiddle + DIT butterflies
    u32 twiddle;    // apply the inter-pass twiddle (0 on the pass with S == 1)
    u32 scale;      // 0 none; 1 load * sc[bitrev_N(i)]; 2 load * sc[i]; 3 store * sc[bitrev_N(i)];
                    // 4 store * sc[i]; 5 store * sc_hi[0] (constant).  Constant factors (1/N, and inside computeH
                    // the 1/N of the preceding inverse transform or den) ride in sc_hi for free
    u32 n_valid;    // loads at global index >= n_valid read as zero (fused zero padding)
    u32 lds_pad;    // strided passes: one unused slot after the C columns of every row (C + 1 slots per row): a wave reading a
                    // column (stride C slots = 64 B at C = 4) then walks all banks instead of four of them
    // Direct factor tables IN THE DATA'S LAYOUT (element g multiplies table[g]; coalesced with the data itself), or null:
    // tw_direct replaces the inter-pass twiddle formed from two tables by one product, sc_direct the coset / 1/N / den scaling.
    // One product per element instead of two, for 32 B of extra traffic per element on a VALU-bound pass.
    const Fr *tw_direct;
    const Fr *sc_direct;
    // computeH's two pointwise steps, fused into the edges of its last transform (both in the data's layout, or null):
    // load_mul: every element is multiplied by load_mul[g] on its way in (only on a pass whose load side has no other factor);
    // store_sub: store_sub[g] is subtracted from every element on its way out, after the scaling
    const Fr *load_mul;
    const Fr *store_sub;
    u32 canon;      // 1 on the last pass of a transform whose output leaves the library (or is h): every element is brought back to [0, p)
};

// ---------------------------------------------------------------- lazy butterflies (Harvey)
// Between the stages of a transform the elements are 256-bit representatives that are NOT reduced below p (field.cuh, "lazy
// arithmetic"): a DIF pass keeps them in [0, 2p), a DIT pass in [0, 4p) (4p < 2^256 for Fr).  One conditional subtraction of 2p per
// butterfly instead of three corrections by p, and the twiddle product drops its final conditional subtraction: ~25 of ~360 vector
// instructions.  w == nullptr: the twiddle is 1.  Passes hand the elements on through HBM as they are (anything below 4p is a valid
// input of a DIT butterfly and of any edge product); only NttPass::canon stores canonical values.
MI_HD void ntt_bfly_dif(Fr &x0, Fr &x1, const Fr *w) {   // x0, x1 in [0, 2p) -> x0 + x1, (x0 - x1) w, both in [0, 2p)
    const Fr u = fe_condsub_2p(fe_add_nored(x0, x1));
    const Fr t = fe_sub_plus2p(x0, x1);                   // (0, 4p)
    x1 = w ? fe_mul_lazy(t, *w) : fe_condsub_2p(t);
    x0 = u;
}
MI_HD void ntt_bfly_dit(Fr &x0, Fr &x1, const Fr *w) {   // x0, x1 in [0, 4p) -> x0 + x1 w, x0 - x1 w, both in [0, 4p)
    const Fr a = fe_condsub_2p(x0);
    const Fr t = w ? fe_mul_lazy(x1, *w) : fe_condsub_2p(x1);   // [0, 2p)
    x0 = fe_add_nored(a, t);
    x1 = fe_sub_plus2p(a, t);
}
// the pointwise product of two lazily reduced elements (both below 4p): below 2p
MI_HD Fr ntt_mul_lazy2(const Fr &x, const Fr &y) { return fe_mul_lazy(fe_condsub_2p(x), fe_condsub_2p(y)); }

MI_HD u32 bitrev_u32(u32 x, u32 bits) {
#if defined(__HIPCC__)
    return bits ? (__builtin_bitreverse32(x) >> (32 - bits)) : 0u;   // v_bfrev_b32
#else
    u32 r = 0;
    for (u32 k = 0; k < bits; k++) { r = (r << 1) | (x & 1); x >>= 1; }
    return r;
#endif
}
MI_HD Fr pow_from_tables(const Fr *lo, const Fr *hi, u32 h, u32 e) {
    return lo[e & ((1u << h) - 1)] * hi[e >> h];
}

// tile-local element index (rho, col) -> global element index, and LDS slot.
// Strided pass (S > 1): tile = R rows x C consecutive columns; LDS slot = rho*C + col.
// Contiguous pass (S == 1): tile = C consecutive blocks of R elements; LDS slot = col*R + rho.
MI_HD u64 ntt_global_index(const NttPass &p, u64 tile, u32 rho, u32 col) {
    if (p.log_s == 0) return ((tile << p.log_c) + col) * ((u64)1 << p.log_r) + rho;
    u64 tiles_per_block = (u64)1 << (p.log_s - p.log_c);
    u64 hi = tile / tiles_per_block, lo0 = (tile % tiles_per_block) << p.log_c;
    return (hi << (p.log_r + p.log_s)) + ((u64)rho << p.log_s) + lo0 + col;
}
MI_HD u32 ntt_lds_slot(const NttPass &p, u32 rho, u32 col) {
    return p.log_s == 0 ? (col << p.log_r) + rho : rho * ((1u << p.log_c) + p.lds_pad) + col;
}
// slots of one plane of the tile's LDS image (the two planes hold the low / high 16 bytes of every element)
MI_HD u32 ntt_plane_slots(const NttPass &p) {
    return p.log_s == 0 ? 1u << (p.log_r + p.log_c) : ((1u << p.log_c) + p.lds_pad) << p.log_r;
}
MI_HD void lds_put(U4 *lds, u32 plane_elems, u32 slot, const Fr &v) {
    lds[slot] = U4{v.l[0], v.l[1], v.l[2], v.l[3]};
    lds[plane_elems + slot] = U4{v.l[4], v.l[5], v.l[6], v.l[7]};
}
MI_HD Fr lds_get(const U4 *lds, u32 plane_elems, u32 slot) {
    U4 a = lds[slot], b = lds[plane_elems + slot];
    Fr v;
    v.l[0] = a.x; v.l[1] = a.y; v.l[2] = a.z; v.l[3] = a.w;
    v.l[4] = b.x; v.l[5] = b.y; v.l[6] = b.z; v.l[7] = b.w;
    return v;
}
// (inter-pass twiddle of (rho, lo): w_M^(lo * bitrev_R(rho)) = w_N^((lo * bitrev_R(rho)) << (log_n - log_m)), see ntt_edge_factor)
// The ONE factor (if any) an element is multiplied by on its way into the tile (phase 0: coset / constant pre-scale, DIT
// pre-twiddle) or out of it (phase 1: DIF post-twiddle, inverse / coset post-scale).  A pass never has a scale and a twiddle on
// the same side (the scales sit on the contiguous pass's load or store, or on the side opposite to the twiddle), so each phase
// has one composing product site and one applying product site -- the product is ~330 instructions and is inlined.
MI_HD bool ntt_edge_factor(const NttPass &p, const NttTables &t, u32 rho, u64 g, u32 phase, Fr &f) {
    const bool sc = phase == 0 ? (p.scale == 1 || p.scale == 2) : p.scale >= 3;
    const bool tw = p.twiddle && (phase == 0) == (p.dit != 0);
    const Fr *lo, *hi;
    u32 e;
    if (sc) {
        if (p.scale == 5) { f = t.sc_hi[0]; return true; }
        if (p.sc_direct) { f = p.sc_direct[g]; return true; }
        e = (p.scale == 1 || p.scale == 3) ? bitrev_u32((u32)g, p.log_n) : (u32)g;
        lo = t.sc_lo; hi = t.sc_hi;
    } else if (tw) {
        const u32 log_m = p.log_r + p.log_s;
        if (p.tw_direct) { f = p.tw_direct[g & (((u64)1 << log_m) - 1)]; return true; }
        const u32 x = (u32)(g & (((u64)1 << p.log_s) - 1)) * bitrev_u32(rho, p.log_r);   // exponent of w_M, x < M
        if (t.tw_64k && log_m <= 16) { f = t.tw_64k[x << (16 - log_m)]; return true; }   // w_M^x = w_65536^(x * 65536/M)
        e = x << (p.log_n - log_m);
        lo = t.tw_lo; hi = t.tw_hi;
    } else if (phase == 0 && p.load_mul) {
        if (g >= p.n_valid) return false;   // zero padding: the element is zero and load_mul may be as short as the data (computeH's c = a o b)
        f = p.load_mul[g];
        return true;
    } else {
        return false;
    }
    f = pow_from_tables(lo, hi, t.tw_h, e);
    return true;
}

// phase 1: global -> LDS (+ fused zero padding, coset pre-scale, DIT pre-twiddle)
MI_HD void ntt_tile_load(const NttPass &p, const NttTables &t, const Fr *data, u64 tile, u32 tid, u32 nthr, U4 *lds) {
    const u32 E = 1u << (p.log_r + p.log_c);
    for (u32 e = tid; e < E; e += nthr) {
        // walk the tile in the order that is contiguous in global memory
        u32 rho, col;
        if (p.log_s == 0) { rho = e & ((1u << p.log_r) - 1); col = e >> p.log_r; }
        else { col = e & ((1u << p.log_c) - 1); rho = e >> p.log_c; }
        u64 g = ntt_global_index(p, tile, rho, col);
        Fr v = g < p.n_valid ? data[g] : Fr::zero(), f;
        if (ntt_edge_factor(p, t, rho, g, 0, f)) v = p.load_mul ? ntt_mul_lazy2(v, f) : fe_mul_lazy(v, f);   // (load_mul: the factor is data too, not a table constant)
        lds_put(lds, ntt_plane_slots(p), ntt_lds_slot(p, rho, col), v);
    }
}
// phase 2: one radix-2 stage over the rows.  stage = 0 .. log_r-1 in execution order.
MI_HD void ntt_tile_stage(const NttPass &p, const NttTables &t, u32 stage, u32 tid, u32 nthr, U4 *lds) {
    const u32 E = 1u << (p.log_r + p.log_c);
    // DIF: half-distance d = R/2, R/4, ..., 1 ; DIT: d = 1, 2, ..., R/2
    const u32 log_d = p.dit ? stage : (p.log_r - 1 - stage);
    const u32 d = 1u << log_d;
    for (u32 b = tid; b < E / 2; b += nthr) {
        // butterfly b -> (column, pair index q within the rows)
        u32 col, q;
        if (p.log_s == 0) { q = b & ((1u << (p.log_r - 1)) - 1); col = b >> (p.log_r - 1); }
        else { col = b & ((1u << p.log_c) - 1); q = b >> p.log_c; }
        u32 j = q & (d - 1);                 // position inside the half block
        u32 r0 = ((q >> log_d) << (log_d + 1)) + j;
        u32 r1 = r0 + d;
        u32 s0 = ntt_lds_slot(p, r0, col), s1 = ntt_lds_slot(p, r1, col);
        const u32 PL = ntt_plane_slots(p);
        Fr x = lds_get(lds, PL, s0), y = lds_get(lds, PL, s1);
        // twiddle w_(2d)^j = w_4096^(j * 2048/d)
        const Fr *w = j ? &t.small[j << (11 - log_d)] : nullptr;
        if (p.dit) ntt_bfly_dit(x, y, w); else ntt_bfly_dif(x, y, w);
        lds_put(lds, PL, s0, x);
        lds_put(lds, PL, s1, y);
    }
}
// phase 3: LDS -> global (+ DIF post-twiddle, inverse / coset post-scale)
MI_HD void ntt_tile_store(const NttPass &p, const NttTables &t, Fr *data, u64 tile, u32 tid, u32 nthr, const U4 *lds) {
    const u32 E = 1u << (p.log_r + p.log_c);
    for (u32 e = tid; e < E; e += nthr) {
        u32 rho, col;
        if (p.log_s == 0) { rho = e & ((1u << p.log_r) - 1); col = e >> p.log_r; }
        else { col = e & ((1u << p.log_c) - 1); rho = e >> p.log_c; }
        u64 g = ntt_global_index(p, tile, rho, col);
        Fr v = lds_get(lds, ntt_plane_slots(p), ntt_lds_slot(p, rho, col)), f;
        if (ntt_edge_factor(p, t, rho, g, 1, f)) v = fe_mul_lazy(v, f);                       // below 2p
        if (p.store_sub) v = fe_sub_plus2p(fe_condsub_2p(v), fe_condsub_2p(p.store_sub[g]));  // below 4p
        if (p.canon) v = fe_canon(v);
        data[g] = v;
    }
}

// ---------------------------------------------------------------- plan: radices of the passes
struct NttPlan {
    u32 n_pass;
    u32 log_r[8];
};
// Contiguous pass up to 2^max_contig, strided passes up to 2^max_strided each.
MI_HD NttPlan ntt_make_plan(u32 log_n, u32 max_contig, u32 max_strided) {
    NttPlan pl;
    pl.n_pass = 0;
    if (log_n <= max_contig) { pl.n_pass = 1; pl.log_r[0] = log_n; return pl; }
    u32 rest = log_n - max_contig;
    u32 ns = (rest + max_strided - 1) / max_strided;
    // spread `rest` evenly over ns strided passes; order = execution order for DIF (strided first)
    for (u32 i = 0; i < ns; i++) {
        u32 r = rest / (ns - i);
        pl.log_r[pl.n_pass++] = r;
        rest -= r;
    }
    pl.log_r[pl.n_pass++] = max_contig;
    return pl;
}
