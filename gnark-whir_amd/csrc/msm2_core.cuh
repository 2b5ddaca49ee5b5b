// Fixed-base ("precomputed") MSM: per-thread bodies shared by the HIP kernels (msm2.hip) and the host emulation.
//
// The proving key is static, so mi_pk_load can store, next to every base P_i, the multiples 2^(c*w) * P_i (w < nwin).
// A scalar's w-th signed digit d then contributes d * (2^(c*w) P_i): ALL windows share ONE set of 2^(c-1) buckets, the
// window combine disappears and c can grow to 22 bits (12 digits per 254-bit scalar instead of 16) because the bucket
// count no longer multiplies by the number of windows.  Same result as gnark-crypto's MultiExp (a canonical group
// element); replaces the same reference functions as msm_core.cuh.
//
//   entry      key = |d| - 1 (21 bits), value = (w * n + i) | sign << 31     -> point = pre[w][i]
//   slices     pass 1 cuts the scalars into slices of <= MSM2_SLICE; a slice's entries fit an LDS staging area
//   sort       two passes: (1) partition by hi = key >> gbits (2^(c-1-gbits) groups, long contiguous runs per slice);
//              (2) inside a group, LDS counting sort by lo = key & (2^gbits - 1) over fixed-size chunks of the group.
//              Narrow groups (gbits = 10..12) shrink the chunk histograms 8..32x and a group's output window to an L2's
//              size (the scatter walks the chunks XCD by XCD, msm.hip), and small chunks (8192) put 13 000 workgroups on
//              the 256 CUs: 0.64 ms for 109 M entries against 1.6 ms with 2^15-bucket groups and 64 K-entry chunks.
//   after that the item / level / bucket-reduce machinery of msm_core.cuh runs unchanged on nkeys = 2^(c-1), one window.
#pragma once
#include "msm_core.cuh"

struct Msm2Shape {
    u32 c;          // window bits (17..22)
    u32 nwin;       // ceil(256 / c)
    u32 nkeys;      // keys of the sort: 2^(c-1) buckets, one set for all windows (fixed-base); nwin * 2^(c-1) when wkeys
    u32 half;       // 2^(c-1): buckets of one window
    u32 wkeys;      // 1: the GENERIC MSM on arbitrary bases borrows this sort (msm.hip): key = (w << (c-1)) | bucket, value =
                    //    point index | sign -- no window copies exist, every window keeps its own bucket set
    u32 gbits;      // log2 of the buckets per group (<= 15: the group-local bucket travels as u16)
    u32 gsize;      // 1 << gbits
    u32 ngroups;    // nkeys >> gbits
    u32 nslices;    // slices of the scalar range for pass 1
    u32 chunk;      // entries per pass-2 chunk
    u32 n;          // pairs
};
MI_HD Msm2Shape msm2_shape(u32 n, u32 c, u32 nslices, u32 chunk, u32 gbits = 15, u32 wkeys = 0) {
    Msm2Shape s;
    s.c = c; s.nwin = (256 + c - 1) / c; s.half = 1u << (c - 1); s.wkeys = wkeys; s.nkeys = wkeys ? s.nwin * s.half : s.half;
    s.gbits = gbits; s.gsize = 1u << gbits; s.ngroups = s.nkeys >> gbits;
    s.nslices = nslices; s.chunk = chunk; s.n = n;
    return s;
}
static constexpr u32 MSM2_SLICE = 512;   // scalars per pass-1 workgroup: 512 * 15 windows * 6 B = 45 KiB of staging
MI_HD void msm2_slice_range(const Msm2Shape &s, u32 g, u32 &begin, u32 &end) {
    u32 per = (s.n + s.nslices - 1) / s.nslices;
    begin = g * per < s.n ? g * per : s.n;
    end = begin + per < s.n ? begin + per : s.n;
}
// signed digit w of the canonical scalar v, carry chained from window 0 (digits in [-2^(c-1), 2^(c-1) - 1])
struct Msm2Digits {
    Fr v;
    u32 carry;
    u32 w;
    MI_HD void start(const Fr &scalar, bool montgomery) { v = montgomery ? fe_from_mont(scalar) : scalar; carry = 0; w = 0; }
    // The low c bits, then the whole number moves down by c (c < 32): indexing v.l[] by the window's bit offset instead put v into
    // scratch memory on the GPU (two scratch loads and a full wait per digit: the counting and partition kernels' waves were parked
    // three quarters of the time).
    MI_HD int32_t next(const Msm2Shape &s) {
        const u32 raw = v.l[0] & ((1u << s.c) - 1);
        for (int k = 0; k < 7; k++) v.l[k] = (v.l[k] >> s.c) | (v.l[k + 1] << (32 - s.c));
        v.l[7] >>= s.c;
        int32_t d = (int32_t)(raw + carry);
        if ((u32)d >= s.half) { d -= (int32_t)(1u << s.c); carry = 1; } else carry = 0;
        w++;
        return d;
    }
    // Window width known at compile time (C = 16..22; C = 0: the run-time walk above).  Called from a fully unrolled loop over w: the
    // limb index is then a constant too, a digit is two shifts, an or and a mask of registers, and nothing moves.
    template <u32 C> MI_HD int32_t at(u32 win, const Msm2Shape &s) {
        if (C == 0) return next(s);
        const u32 bit = win * C, limb = bit >> 5, sh = bit & 31;
        u32 raw = limb < 8 ? v.l[limb < 8 ? limb : 0] >> sh : 0u;
        if (sh + C > 32 && limb + 1 < 8) raw |= v.l[limb + 1 < 8 ? limb + 1 : 0] << ((32 - sh) & 31);
        raw &= (1u << (C ? C : 1)) - 1;
        int32_t d = (int32_t)(raw + carry);
        if ((u32)d >= s.half) { d -= (int32_t)(1u << (C ? C : 1)); carry = 1; } else carry = 0;
        return d;
    }
};
// windows of a width known at compile time (C = 0: s.nwin)
template <u32 C> MI_HD u32 msm2_nwin_c(const Msm2Shape &s) { return C ? (256 + (C ? C : 1) - 1) / (C ? C : 1) : s.nwin; }
// CALL(C) with the compile-time width that equals c, or CALL(0) (run-time width) for any other
#define MSM2_FOR_C(c, CALL) \
    switch (c) { case 16: CALL(16); break; case 17: CALL(17); break; case 18: CALL(18); break; case 19: CALL(19); break; \
                 case 20: CALL(20); break; case 21: CALL(21); break; case 22: CALL(22); break; default: CALL(0); break; }

// ---- pass 1a: workgroup = slice g; LDS hist[ngroups]; C1[hi][g]
// one scalar (already canonical in dg, restarted by the caller): its non-zero digits counted by group
template <u32 C = 0>
MI_HD void msm2_count_one(const Msm2Shape &s, Msm2Digits dg, u32 *lds) {
    const u32 nw = msm2_nwin_c<C>(s);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (u32 w = 0; w < nw; w++) {
        int32_t d = dg.template at<C>(w, s);
        if (d) MI_LDS_ATOMIC_ADD(&lds[(((u32)(d < 0 ? -d : d) - 1) + (s.wkeys ? w * s.half : 0u)) >> s.gbits], 1u);
    }
}
MI_HD void msm2_count_body(const Msm2Shape &s, const Fr *scalars, bool montgomery, u32 g, u32 *lds, u32 tid, u32 nthr) {
    u32 begin, end;
    msm2_slice_range(s, g, begin, end);
    for (u32 i = begin + tid; i < end; i += nthr) {
        Msm2Digits dg;
        dg.start(scalars[i], montgomery);
        msm2_count_one(s, dg, lds);
    }
}
// The same for `cnt` consecutive slices g0, g0 + 1, ... (counters of slice j at lds + j * ngroups) with the NEXT scalar's load in flight
// while the current one is counted: the counting workgroups of msm.hip walk 32 slices each, and one dependent load per slice left
// their waves parked three quarters of the time.
template <u32 C = 0>
MI_HD void msm2_count_slices(const Msm2Shape &s, const Fr *scalars, bool montgomery, u32 g0, u32 cnt, u32 *lds, u32 tid, u32 nthr) {
    u32 per = (s.n + s.nslices - 1) / s.nslices;
    if (per > nthr) {   // (not the shape msm.hip launches: MSM2_SLICE scalars per slice, as many threads)
        for (u32 j = 0; j < cnt; j++) msm2_count_body(s, scalars, montgomery, g0 + j, lds + j * s.ngroups, tid, nthr);
        return;
    }
    u32 begin, end;
    msm2_slice_range(s, g0, begin, end);
    bool have = begin + tid < end;
    Fr next = have ? scalars[begin + tid] : Fr{};
    for (u32 j = 0; j < cnt; j++) {
        const Fr cur = next;
        const bool have_cur = have;
        if (j + 1 < cnt) {
            msm2_slice_range(s, g0 + j + 1, begin, end);
            have = begin + tid < end;
            if (have) next = scalars[begin + tid];
        }
        if (have_cur) {
            Msm2Digits dg;
            dg.start(cur, montgomery);
            msm2_count_one<C>(s, dg, lds + j * s.ngroups);
        }
    }
}
// ---- pass 1b: partition, staged through LDS so that the global stores are runs, not single entries.
// The workgroup of slice g has hist[hi] (msm2_count_body again), loff = exclusive scan of hist (loff[ngroups] = entries
// of the slice), cursor[hi] = loff[hi].  place: every entry goes to its group's run inside the LDS staging area;
// copy: entry e of the staging area belongs to the group hi with loff[hi] <= e < loff[hi+1] and lands at
// gbase[hi] + (e - loff[hi]), gbase[hi] = S1[hi * G + g]: consecutive e of a group are consecutive in part_lo / part_val.
// one scalar i (canonical in dg): its entries go to their groups' runs of the staging area.  stage_grp keeps every entry's group
// next to it: the copy phase then needs two LDS reads per entry instead of a binary search over loff (8 dependent reads).
template <u32 C = 0>
MI_HD void msm2_place_one(const Msm2Shape &s, Msm2Digits dg, u32 i, u32 *cursor, uint16_t *stage_lo, u32 *stage_val, uint16_t *stage_grp) {
    const u32 nw = msm2_nwin_c<C>(s);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (u32 w = 0; w < nw; w++) {
        int32_t d = dg.template at<C>(w, s);
        if (!d) continue;
        u32 key = (u32)(d < 0 ? -d : d) - 1 + (s.wkeys ? w * s.half : 0u);
        u32 pos = MI_LDS_ATOMIC_ADD(&cursor[key >> s.gbits], 1u);
        stage_lo[pos] = (uint16_t)(key & (s.gsize - 1));
        stage_grp[pos] = (uint16_t)(key >> s.gbits);
        stage_val[pos] = (s.wkeys ? i : w * s.n + i) | (d < 0 ? 0x80000000u : 0u);
    }
}
template <u32 C = 0>
MI_HD void msm2_stage_place_body(const Msm2Shape &s, const Fr *scalars, bool montgomery, u32 g, u32 *cursor, uint16_t *stage_lo, u32 *stage_val,
                                 uint16_t *stage_grp, u32 tid, u32 nthr) {
    u32 begin, end;
    msm2_slice_range(s, g, begin, end);
    for (u32 i = begin + tid; i < end; i += nthr) {
        Msm2Digits dg;
        dg.start(scalars[i], montgomery);
        msm2_place_one<C>(s, dg, i, cursor, stage_lo, stage_val, stage_grp);
    }
}
// gbase[hi] = S1[hi * G + g] (the slice's run of group hi in the partitioned arrays), loaded once per workgroup
MI_HD void msm2_stage_copy_body(const Msm2Shape &s, const u32 *gbase, const u32 *loff, const uint16_t *stage_lo, const u32 *stage_val,
                                const uint16_t *stage_grp, uint16_t *part_lo, u32 *part_val, u32 tid, u32 nthr) {
    const u32 total = loff[s.ngroups];
    for (u32 e = tid; e < total; e += nthr) {
        const u32 grp = stage_grp[e];
        u32 pos = gbase[grp] + (e - loff[grp]);
        part_lo[pos] = stage_lo[e];
        part_val[pos] = stage_val[e];
    }
}
// ---- chunk table: group hi owns entries [gstart[hi], gstart[hi+1]); it is cut into ceil(size / chunk) chunks.
// thread = group: gstart[hi] and the group's chunk count; cstart = exclusive scan of the counts (cstart[ngroups] = total chunks)
MI_HD void msm2_chunk_count_body(const Msm2Shape &s, const u32 *S1, u32 *gstart, u32 *nchunks, u32 hi) {
    u32 a = S1[(size_t)hi * s.nslices], b = S1[(size_t)(hi + 1) * s.nslices];   // S1 has ngroups*G + 1 entries
    gstart[hi] = a;
    nchunks[hi] = (b - a + s.chunk - 1) / s.chunk;
    if (hi + 1 == s.ngroups) gstart[s.ngroups] = b;
}
// chunk id -> (group, entry range)
MI_HD bool msm2_chunk_range(const Msm2Shape &s, const u32 *gstart, const u32 *cstart, u32 chunk_id, u32 &hi, u32 &b, u32 &e) {
    if (chunk_id >= cstart[s.ngroups]) return false;
    u32 lo_g = 0, hi_g = s.ngroups;   // largest group with cstart <= chunk_id (groups without chunks share a start with their successor)
    while (hi_g - lo_g > 1) {
        u32 mid = (lo_g + hi_g) >> 1;
        if (cstart[mid] <= chunk_id) lo_g = mid; else hi_g = mid;
    }
    hi = lo_g;
    b = gstart[hi] + (chunk_id - cstart[hi]) * s.chunk;
    e = gstart[hi + 1];
    if (b + s.chunk < e) e = b + s.chunk;
    return true;
}
// ---- pass 2a: workgroup = chunk; LDS hist[gsize] (u32); H2[chunk][lo]
MI_HD void msm2_hist2_zero(const Msm2Shape &s, u32 *lds, u32 tid, u32 nthr) {
    for (u32 b = tid; b < s.gsize; b += nthr) lds[b] = 0;
}
// (eight entries per trip here and in the scatter: their loads go out together instead of one exposed latency per entry)
MI_HD void msm2_hist2_count(const uint16_t *part_lo, u32 b, u32 e, u32 *lds, u32 tid, u32 nthr) {
    u32 k = b + tid;
    for (; k + 7 * nthr < e; k += 8 * nthr) {
        uint16_t lo[8];
        for (u32 j = 0; j < 8; j++) lo[j] = part_lo[k + j * nthr];
        for (u32 j = 0; j < 8; j++) MI_LDS_ATOMIC_ADD(&lds[lo[j]], 1u);
    }
    for (; k < e; k += nthr) MI_LDS_ATOMIC_ADD(&lds[part_lo[k]], 1u);
}
MI_HD void msm2_hist2_write(const Msm2Shape &s, u32 *H2, u32 chunk_id, const u32 *lds, u32 tid, u32 nthr) {
    for (u32 b = tid; b < s.gsize; b += nthr) H2[(size_t)chunk_id * s.gsize + b] = lds[b];
}
// ---- column sums: thread = key (hi, lo): H2[chunk][lo] <- exclusive prefix along the chunks of group hi (in place),
// total[key] <- number of entries of the key
// Eight chunks per trip: the eight loads go out together (a group that holds the WHIR mix's small scalars has hundreds of chunks, and one
// dependent load + store per chunk made that group's threads the launch's tail: 0.24 ms for 35 M entries).
MI_HD u32 msm2_colsum_body(const Msm2Shape &s, const u32 *cstart, u32 *H2, u32 *total, u32 key) {
    u32 hi = key >> s.gbits, lo = key & (s.gsize - 1), run = 0;
    u32 ch = cstart[hi];
    const u32 end = cstart[hi + 1];
    for (; ch + 8 <= end; ch += 8) {
        u32 v[8];
        for (u32 k = 0; k < 8; k++) v[k] = H2[(size_t)(ch + k) * s.gsize + lo];
        for (u32 k = 0; k < 8; k++) { H2[(size_t)(ch + k) * s.gsize + lo] = run; run += v[k]; }
    }
    for (; ch < end; ch++) {
        size_t i = (size_t)ch * s.gsize + lo;
        u32 v = H2[i];
        H2[i] = run;
        run += v;
    }
    total[key] = run;
    return run;
}
// ---- pass 2b: scatter.  cursor[lo] = keystart[hi*gsize + lo] + H2x[chunk][lo]
MI_HD void msm2_scatter2_init(const Msm2Shape &s, const u32 *keystart, const u32 *H2x, u32 chunk_id, u32 hi, u32 *lds, u32 tid, u32 nthr) {
    for (u32 b = tid; b < s.gsize; b += nthr) lds[b] = keystart[(size_t)hi * s.gsize + b] + H2x[(size_t)chunk_id * s.gsize + b];
}
MI_HD void msm2_scatter2_move(const uint16_t *part_lo, const u32 *part_val, u32 b, u32 e, u32 *lds, u32 *sorted, u32 tid, u32 nthr) {
    u32 k = b + tid;
    for (; k + 7 * nthr < e; k += 8 * nthr) {
        uint16_t lo[8];
        u32 val[8], pos[8];
        for (u32 j = 0; j < 8; j++) { lo[j] = part_lo[k + j * nthr]; val[j] = part_val[k + j * nthr]; }
        for (u32 j = 0; j < 8; j++) pos[j] = MI_LDS_ATOMIC_ADD(&lds[lo[j]], 1u);
        for (u32 j = 0; j < 8; j++) sorted[pos[j]] = val[j];
    }
    for (; k < e; k += nthr) {
        u32 pos = MI_LDS_ATOMIC_ADD(&lds[part_lo[k]], 1u);
        sorted[pos] = part_val[k];
    }
}

// ---- pass 2b, staged: the chunk is counting-sorted INSIDE the workgroup first, then leaves in destination order.  A wave's 64 stores of
// the plain scatter above go to 64 different lines (the entries of a chunk arrive in scalar order); in destination order consecutive
// lanes write the consecutive words of a bucket's run (about four at 2^23 uniform pairs, hundreds for the WHIR mix's small scalars), so a
// store instruction touches a quarter of the lines or fewer -- the address path of those stores, not LDS, was what the scatter waited on
// (SQ_WAIT_INST_ANY 0.67, SQ_WAIT_INST_LDS 0.00).
//   rank:  thread t holds entries b + t + j * nthr (j < MSM2_STAGE_PER); rank[j] = its place among the chunk's entries of the same bucket
//   scan:  loc = exclusive scan of the per-bucket counts (the caller's block scan)
//   place: entry -> stage[loc[bucket] + rank]
//   copy:  stage slot p of bucket q goes to sorted[cursor[q] + p - loc[q]]   (cursor as in msm2_scatter2_init)
static constexpr u32 MSM2_STAGE_PER = 8;
MI_HD u32 msm2_stage2_rank(const uint16_t *part_lo, const u32 *part_val, u32 b, u32 e, u32 *cnt, u32 tid, u32 nthr, uint16_t *lo, u32 *val, u32 *rank) {
    u32 m = 0;
    for (u32 j = 0; j < MSM2_STAGE_PER; j++) {
        const u32 k = b + tid + j * nthr;
        if (k < e) { lo[j] = part_lo[k]; val[j] = part_val[k]; m = j + 1; }
    }
    for (u32 j = 0; j < MSM2_STAGE_PER; j++) if (j < m) rank[j] = MI_LDS_ATOMIC_ADD(&cnt[lo[j]], 1u);
    return m;
}
MI_HD void msm2_stage2_place(const u32 *loc, const uint16_t *lo, const u32 *val, const u32 *rank, u32 m, uint16_t *st_lo, u32 *st_val) {
    for (u32 j = 0; j < MSM2_STAGE_PER; j++) if (j < m) {
        const u32 p = loc[lo[j]] + rank[j];
        st_lo[p] = lo[j];
        st_val[p] = val[j];
    }
}
MI_HD void msm2_stage2_copy(const u32 *cursor, const u32 *loc, const uint16_t *st_lo, const u32 *st_val, u32 n, u32 *sorted, u32 tid, u32 nthr) {
    for (u32 p = tid; p < n; p += nthr) {
        const u32 q = st_lo[p];
        sorted[cursor[q] + (p - loc[q])] = st_val[p];
    }
}

// ---- pk_load: the window copies pre[w][i] = 2^(c*w) * P_i, affine.  Thread i walks the windows.
template <class F>
MI_HD void msm2_precompute_body(const Affine<F> *base, Affine<F> *pre, u32 n, u32 c, u32 nwin, u32 i) {
    Affine<F> p = base[i];
    pre[i] = p;
    XYZZ<F> acc = XYZZ<F>::from_affine(p);
    for (u32 w = 1; w < nwin; w++) {
        for (u32 k = 0; k < c; k++) acc = xyzz_dbl(acc);
        Affine<F> a = xyzz_to_affine(acc);
        pre[(size_t)w * n + i] = a;
        acc = XYZZ<F>::from_affine(a);   // keep the chain in affine-normalised form (shorter numbers of squarings are not needed)
    }
}
