// Pippenger multi-scalar multiplication, per-thread bodies shared by the HIP kernels (msm.hip)
// and the host emulation of the CPU tests.
//
// Replaces gnark-crypto ecc/bn254 G1Jac.MultiExp / G2Jac.MultiExp (+ partitionScalars) on the path
// reached from /root/reference/mt.go:496.  Same mathematical result
// (sum_i s_i * P_i, a canonical group element), different organisation:
//
//   1 digits   : scalar -> canonical -> nwin signed c-bit digits in [-2^(c-1), 2^(c-1)-1] (int16,
//                window-major).  key = w * 2^(c-1) + |d| - 1 names a bucket.
//   2 sort     : counting sort of (key -> point index | sign<<31): per-workgroup LDS histograms
//                H[window][slice][bucket], column sums + one exclusive scan over the keys, LDS cursors for the scatter.  No global
//                atomics; the order inside a bucket is irrelevant (the sum is canonical).
//   3 accumulate, level 1: every bucket is cut into items of <= L entries; one thread sums one item
//                with mixed XYZZ additions (uniform work per thread, whatever the skew of the scalars
//                -- the WHIR witness is ~45% {0,1}).  Buckets that fit one item are
//                final; the others leave one partial per item.
//      levels 2..: the same item decomposition over the partials (full XYZZ additions) until every
//                bucket holds one point.
//   4 bucket reduce: per window sum_b (b+1) * bucket[b] by segment running sums, then the same
//                item/level machinery keyed by window.
//   5 the nwin window sums are combined (Horner, c doublings per window) on the host: O(nwin)
//                serial point operations, the same place gnark does them.
#pragma once
#include "curve.cuh"

#if defined(__HIP_DEVICE_COMPILE__)
#define MI_LDS_ATOMIC_ADD(p, v) atomicAdd((p), (v))
#else
static inline u32 mi_host_fetch_add(u32 *p, u32 v) { u32 o = *p; *p = o + v; return o; }
#define MI_LDS_ATOMIC_ADD(p, v) mi_host_fetch_add((p), (v))
#endif

struct MsmShape {
    u32 c;          // window bits, 2..16
    u32 nwin;       // ceil(256 / c): the top window then never carries out (scalars < 2^254)
    u32 nbuckets;   // 2^(c-1) per window
    u32 nkeys;      // nwin * nbuckets
    u32 nslices;    // G: slices of the scalar range, one histogram workgroup per (slice, window)
    u32 n;          // number of (point, scalar) pairs
};
MI_HD MsmShape msm_shape(u32 n, u32 c, u32 nslices) {
    MsmShape s;
    s.c = c; s.nwin = (256 + c - 1) / c; s.nbuckets = 1u << (c - 1); s.nkeys = s.nwin * s.nbuckets;
    s.nslices = nslices; s.n = n;
    return s;
}

// ---- 1. digits: thread i.  digits laid out [w][i] (window-major) as int16
MI_HD void msm_digits_body(const MsmShape &s, const Fr *scalars, bool montgomery, int16_t *digits, u32 i) {
    Fr v = scalars[i];
    if (montgomery) v = fe_from_mont(v);
    u32 carry = 0;
    const u32 mask = (1u << s.c) - 1, half = s.nbuckets;
    for (u32 w = 0; w < s.nwin; w++) {
        u32 bit = w * s.c, limb = bit >> 5, sh = bit & 31;
        u64 lo = limb < 8 ? v.l[limb] : 0u, hi = limb + 1 < 8 ? v.l[limb + 1] : 0u;
        u32 raw = (u32)(((hi << 32) | lo) >> sh) & mask;
        int32_t d = (int32_t)(raw + carry);
        if ((u32)d >= half) { d -= (int32_t)(1u << s.c); carry = 1; } else carry = 0;
        digits[(size_t)w * s.n + i] = (int16_t)d;
    }
}

// slice g of the scalar range
MI_HD void msm_slice_range(const MsmShape &s, u32 g, u32 &begin, u32 &end) {
    u32 per = (s.n + s.nslices - 1) / s.nslices;
    begin = g * per < s.n ? g * per : s.n;
    end = begin + per < s.n ? begin + per : s.n;
}

// ---- 2a. histogram: workgroup (g, w), LDS hist[nbuckets]; three phases separated by barriers
MI_HD void msm_hist_zero(const MsmShape &s, u32 *lds, u32 tid, u32 nthr) {
    for (u32 b = tid; b < s.nbuckets; b += nthr) lds[b] = 0;
}
MI_HD void msm_hist_count(const MsmShape &s, const int16_t *digits, u32 g, u32 w, u32 *lds, u32 tid, u32 nthr) {
    u32 begin, end;
    msm_slice_range(s, g, begin, end);
    const int16_t *dw = digits + (size_t)w * s.n;
    for (u32 i = begin + tid; i < end; i += nthr) {
        int32_t d = dw[i];
        if (d) MI_LDS_ATOMIC_ADD(&lds[(d < 0 ? -d : d) - 1], 1u);
    }
}
// H[w][g][b] (bucket fastest): every workgroup writes / reads whole contiguous rows
MI_HD size_t msm_cell(const MsmShape &s, u32 w, u32 g, u32 b) { return ((size_t)w * s.nslices + g) * s.nbuckets + b; }
MI_HD void msm_hist_write(const MsmShape &s, u32 *H, u32 g, u32 w, const u32 *lds, u32 tid, u32 nthr) {
    for (u32 b = tid; b < s.nbuckets; b += nthr) H[msm_cell(s, w, g, b)] = lds[b];
}
// column sums: thread = key (w, b).  H[w][g][b] <- sum_{g' < g} H[w][g'][b] (exclusive along the slices),
// total[key] <- sum over all slices.  Neighbouring threads touch neighbouring words for every g.
MI_HD void msm_colsum_body(const MsmShape &s, u32 *H, u32 *total, u32 key) {
    u32 w = key / s.nbuckets, b = key % s.nbuckets, run = 0;
    for (u32 g = 0; g < s.nslices; g++) {
        size_t i = msm_cell(s, w, g, b);
        u32 v = H[i];
        H[i] = run;
        run += v;
    }
    total[key] = run;
}
// ---- 2b. scatter: workgroup (g, w), LDS cursor[b] = keystart[key] + entries of earlier slices
MI_HD void msm_scatter_init(const MsmShape &s, const u32 *keystart, const u32 *Hx, u32 g, u32 w, u32 *lds, u32 tid, u32 nthr) {
    for (u32 b = tid; b < s.nbuckets; b += nthr) lds[b] = keystart[(size_t)w * s.nbuckets + b] + Hx[msm_cell(s, w, g, b)];
}
MI_HD void msm_scatter_move(const MsmShape &s, const int16_t *digits, u32 g, u32 w, u32 *lds, u32 *sorted, u32 tid, u32 nthr) {
    u32 begin, end;
    msm_slice_range(s, g, begin, end);
    const int16_t *dw = digits + (size_t)w * s.n;
    for (u32 i = begin + tid; i < end; i += nthr) {
        int32_t d = dw[i];
        if (!d) continue;
        u32 pos = MI_LDS_ATOMIC_ADD(&lds[(d < 0 ? -d : d) - 1], 1u);
        sorted[pos] = i | (d < 0 ? 0x80000000u : 0u);
    }
}

// ---- 3. item decomposition.  Per key: start (first entry), cnt (entries), items = ceil(cnt / L).
// level 1 prep from keystart = exclusive scan of the per-key totals (nkeys + 1 entries, last = grand total)
MI_HD void msm_prep_level1(const MsmShape &s, const u32 *keystart, u32 L, u32 *start, u32 *cnt, u32 *items, u32 key) {
    u32 a = keystart[key], b = keystart[key + 1];
    start[key] = a; cnt[key] = b - a; items[key] = (b - a + L - 1) / L;
}
// level k+1 prep from level k: keys that produced more than one item continue with their partials
MI_HD void msm_prep_next(const u32 *prev_items, const u32 *prev_item_start, u32 L, u32 *start, u32 *cnt, u32 *items, u32 key) {
    u32 m = prev_items[key];
    m = m > 1 ? m : 0;
    start[key] = prev_item_start[key]; cnt[key] = m; items[key] = (m + L - 1) / L;
}
// key of an item: the largest key with item_start[key] <= item  (item_start = exclusive scan of items)
MI_HD u32 msm_item_key(const u32 *item_start, u32 nkeys, u32 item) {
    u32 lo = 0, hi = nkeys;  // invariant: item_start[lo] <= item, answer in [lo, hi)
    while (hi - lo > 1) {
        u32 mid = (lo + hi) >> 1;
        if (item_start[mid] <= item) lo = mid; else hi = mid;
    }
    return lo;
}
// entries [b, e) of item j of a key: the key's cnt entries are cut into `nitems` items whose sizes differ by at most one
// (35 entries at L = 32 become 18 + 17, not 32 + 3: neighbouring lanes then run similar trip counts)
MI_HD void msm_item_range(u32 start, u32 cnt, u32 nitems, u32 j, u32 &b, u32 &e) {
    u32 q = cnt / nitems, r = cnt % nitems;
    b = start + j * q + (j < r ? j : r);
    e = b + q + (j < r ? 1 : 0);
}
template <class F>
MI_HD void msm_accum_affine_body(const Affine<F> *pts, const u32 *sorted, const u32 *start, const u32 *cnt, const u32 *items,
                                 const u32 *item_start, u32 nkeys, u32 L, XYZZ<F> *bucket, XYZZ<F> *partial_out, u32 item) {
    if (item >= item_start[nkeys]) return;
    u32 key = msm_item_key(item_start, nkeys, item);
    u32 b, e;
    msm_item_range(start[key], cnt[key], items[key], item - item_start[key], b, e);
    XYZZ<F> acc = XYZZ<F>::inf();
    for (u32 k = b; k < e; k++) {
        u32 v = sorted[k];
        Affine<F> p = pts[v & 0x7fffffffu];
        xyzz_madd(acc, p, (v >> 31) != 0);
    }
    if (items[key] == 1) bucket[key] = acc; else partial_out[item] = acc;
}
template <class F>
MI_HD void msm_accum_xyzz_body(const XYZZ<F> *partial_in, const u32 *start, const u32 *cnt, const u32 *items, const u32 *item_start,
                               u32 nkeys, u32 L, XYZZ<F> *bucket, XYZZ<F> *partial_out, u32 item) {
    if (item >= item_start[nkeys]) return;
    u32 key = msm_item_key(item_start, nkeys, item);
    u32 b, e;
    msm_item_range(start[key], cnt[key], items[key], item - item_start[key], b, e);
    XYZZ<F> acc = partial_in[b];
    for (u32 k = b + 1; k < e; k++) xyzz_add(acc, partial_in[k]);
    if (items[key] == 1) bucket[key] = acc; else partial_out[item] = acc;
}

// ---- 4. bucket reduce: thread t of window w owns buckets [t*seg, (t+1)*seg) (weights b+1):
//   out = sum_j (base + j + 1) * B[base + j] = acc + base * run,  base = t*seg
template <class F>
MI_HD void msm_bucket_reduce_body(const XYZZ<F> *bucket, u32 nbuckets, u32 seg, XYZZ<F> *out, u32 w, u32 t) {
    u32 base = t * seg;
    const XYZZ<F> *B = bucket + (size_t)w * nbuckets + base;
    u32 m = base + seg <= nbuckets ? seg : nbuckets - base;
    XYZZ<F> run = XYZZ<F>::inf(), acc = XYZZ<F>::inf();
    for (u32 j = m; j-- > 0;) {
        xyzz_add(run, B[j]);
        xyzz_add(acc, run);
    }
    if (base) {
        XYZZ<F> sc = xyzz_mul_u32(run, base);
        xyzz_add(acc, sc);
    }
    out[(size_t)w * ((nbuckets + seg - 1) / seg) + t] = acc;
}

// ---- 5. host: total = sum_w 2^(c*w) * window_sum[w]
template <class F>
MI_HD XYZZ<F> msm_combine_windows(const XYZZ<F> *wsum, u32 nwin, u32 c) {
    XYZZ<F> tot = XYZZ<F>::inf();
    for (u32 w = nwin; w-- > 0;) {
        for (u32 k = 0; k < c; k++) tot = xyzz_dbl(tot);
        xyzz_add(tot, wsum[w]);
    }
    return tot;
}
