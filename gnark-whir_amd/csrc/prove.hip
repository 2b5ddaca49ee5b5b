// Proving key residency and the fused prove path: everything groth16.Prove does after the solve
// (gnark v0.11.0 backend/groth16/bn254/prove.go, reached from /root/reference/mt.go:496; SURVEY.md
// section 3.3 steps 4-8, 8a rows a7 and a9).
//
//   mi_pk_load   uploads pk.G1.{A,B,K,Z}, pk.G2.B once and turns the static masks pk.InfinityA/B and the
//                public / committed wire sets into device gather-index arrays (row a7).
//   prove        computeH on the device -> three gathers of W -> five MSMs -> O(1)-point blinding and
//                assembly on the host (Ar, Bs1, Krs, Bs exactly as prove.go composes them).
#include "prove_internal.h"
#include <cstring>
#include <vector>
#include <chrono>
#include <cstdlib>

__global__ void k_expand_points(G1Aff *full, const G1Aff *compact, const u32 *idx, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) full[idx[i]] = compact[i];
}
__global__ void k_gather_fr(Fr *out, const Fr *W, const u32 *idx, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = W[idx[i]];
}

static int32_t upload(mi_ctx *ctx, void **dst, const void *src, size_t bytes) {
    MI_CHECK_HIP(ctx, hipMalloc(dst, bytes ? bytes : 32));
    if (bytes) MI_CHECK_HIP(ctx, hipMemcpyAsync(*dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    return MI_OK;
}

int32_t mi_pk_load_range(mi_ctx *ctx, const mi_pk_desc *d, mi_pk **out, bool device_points, const ShardRange *sr, bool adopt, bool *took_arrays) {
    if (took_arrays) *took_arrays = false;
    if (!ctx || !d || !out) return MI_EINVAL;
    *out = nullptr;
    // device_points with a range (mi_pk_load_sharded_dev): the arrays ARE this part's slices (counts = points of the slice)
    if (d->log_n > 28 || d->nb_public > d->nb_wires || !d->infinity_a || !d->infinity_b) MI_FAIL(ctx, MI_EINVAL, "pk: bad header");
    const u64 N = (u64)1 << d->log_n;
    const bool slices = sr && device_points;
    if (!slices && d->n_g1_z + 1 < N) MI_FAIL(ctx, MI_EINVAL, "pk: G1.Z needs at least 2^log_n - 1 points");
    if (d->nb_wires >= ((u64)1 << 31)) MI_FAIL(ctx, MI_EINVAL, "pk: too many wires");
    if (d->n_g2_b != d->n_g1_b) MI_FAIL(ctx, MI_EINVAL, "pk: G1.B and G2.B differ in length");
    // gather indices from the static masks (prove.go: wireValuesA/B filters; K drops public + committed)
    // a part of a sharded key keeps the wires [w_lo, w_hi) (indices relative to w_lo) and the points those wires own:
    // a0 / b0 / k0 = points of earlier wires = offset of this part's slice in the caller's arrays
    const u64 w_lo = sr ? sr->w_lo : 0, w_hi = sr ? sr->w_hi : d->nb_wires;
    const u64 z_lo = sr ? sr->z_lo : 0, z_hi = sr ? sr->z_hi : N - 1;
    if (w_lo > w_hi || w_hi > d->nb_wires || z_lo > z_hi || z_hi > N - 1) MI_FAIL(ctx, MI_EINVAL, "pk: bad shard range");
    std::vector<u32> ia, ib, ik;
    u64 ci = 0, ca = 0, cb = 0, ck = 0, a0 = 0, b0 = 0, k0 = 0;
    for (u64 j = 0; j < d->nb_wires; j++) {
        const bool in = j >= w_lo && j < w_hi;
        if (j == w_lo) { a0 = ca; b0 = cb; k0 = ck; }
        if (!d->infinity_a[j]) { ca++; if (in) ia.push_back((u32)(j - w_lo)); }
        if (!d->infinity_b[j]) { cb++; if (in) ib.push_back((u32)(j - w_lo)); }
        if (j >= d->nb_public) {
            while (ci < d->n_committed && d->committed_wires[ci] < j) ci++;
            if (ci < d->n_committed && d->committed_wires[ci] == j) continue;
            ck++;
            if (in) ik.push_back((u32)(j - w_lo));
        }
    }
    if (w_lo >= d->nb_wires) { a0 = ca; b0 = cb; k0 = ck; }
    if (slices) {
        if (ia.size() != d->n_g1_a || ib.size() != d->n_g1_b || ik.size() != d->n_g1_k || d->n_g1_z < z_hi - z_lo)
            MI_FAIL(ctx, MI_EINVAL, "pk: slice point counts do not match the infinity masks / public / committed wire sets of this rank's wire range");
    } else if (ca != d->n_g1_a || cb != d->n_g1_b || ck != d->n_g1_k)
        MI_FAIL(ctx, MI_EINVAL, "pk: point counts do not match the infinity masks / public / committed wire sets");
    mi_pk *pk = new (std::nothrow) mi_pk();
    if (!pk) return MI_ENOMEM;
    pk->log_n = d->log_n; pk->nb_wires = w_hi - w_lo;
    pk->nb_public = (u32)(d->nb_public <= w_lo ? 0 : (d->nb_public >= w_hi ? w_hi - w_lo : d->nb_public - w_lo));
    pk->n_a = ia.size(); pk->n_b = ib.size(); pk->n_k = ik.size(); pk->n_z = sr ? z_hi - z_lo : d->n_g1_z;
    pk->wire_lo = w_lo; pk->z_lo = z_lo; pk->n_z_msm = z_hi - z_lo;
    std::memcpy(&pk->alpha1, &d->alpha1, 64); std::memcpy(&pk->beta1, &d->beta1, 64); std::memcpy(&pk->delta1, &d->delta1, 64);
    std::memcpy(&pk->beta2, &d->beta2, 128); std::memcpy(&pk->delta2, &d->delta2, 128);
    int32_t rc = MI_OK;
    if (device_points) {
        pk->g1_a = (void *)d->g1_a; pk->g1_b = (void *)d->g1_b; pk->g1_k = (void *)d->g1_k; pk->g1_z = (void *)d->g1_z; pk->g2_b = (void *)d->g2_b;
        // mi_pk_load_raw hands its converted arrays over (adopt): from here on they are the key's, on success AND on every failure
        // path below (mi_pk_free releases what is still there); *took_arrays tells the caller so
        pk->owns_points = adopt;
        if (adopt && took_arrays) *took_arrays = true;
    } else {
        pk->owns_points = true;
        if (rc == MI_OK) rc = upload(ctx, &pk->g1_a, d->g1_a + a0, pk->n_a * 64);
        if (rc == MI_OK) rc = upload(ctx, &pk->g1_b, d->g1_b + b0, pk->n_b * 64);
        if (rc == MI_OK) rc = upload(ctx, &pk->g1_k, d->g1_k + k0, pk->n_k * 64);
        if (rc == MI_OK) rc = upload(ctx, &pk->g1_z, d->g1_z + z_lo, pk->n_z * 64);
        if (rc == MI_OK) rc = upload(ctx, &pk->g2_b, d->g2_b + b0, pk->n_b * 128);
    }
    if (rc == MI_OK) rc = upload(ctx, (void **)&pk->idx_a, ia.data(), ia.size() * 4);
    if (rc == MI_OK) rc = upload(ctx, (void **)&pk->idx_b, ib.data(), ib.size() * 4);
    if (rc == MI_OK) rc = upload(ctx, (void **)&pk->idx_k, ik.data(), ik.size() * 4);
    // expanded per-wire copies of A and K
    auto expand = [&](G1Aff **full, const void *compact, const u32 *idx, size_t n) -> int32_t {
        const size_t bytes = (size_t)pk->nb_wires * sizeof(G1Aff);
        MI_CHECK_HIP(ctx, hipMalloc((void **)full, bytes ? bytes : 64));
        MI_CHECK_HIP(ctx, hipMemsetAsync(*full, 0, bytes, ctx->stream));
        if (n) hipLaunchKernelGGL(k_expand_points, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, *full, (const G1Aff *)compact, idx, n);
        MI_CHECK_HIP(ctx, hipGetLastError());
        return MI_OK;
    };
    if (rc == MI_OK) rc = expand(&pk->a_full, pk->g1_a, pk->idx_a, pk->n_a);
    if (rc == MI_OK) rc = expand(&pk->k_full, pk->g1_k, pk->idx_k, pk->n_k);
    // Fixed-base tables.  Measured at N = 2^23 with proofs overlapping: c = 19 / 17 / 20 for A+K / B / Z (round 2: B went
    // from 18 to 17 when the G2 additions got 18 % cheaper and the 2^17-bucket G2 reduce weighed more: 30.2 vs 29.9 proofs/s) gives
    // +7 % proofs/s over the generic c = 16 path (13..15 windows instead of 16); wider windows lose it again to the bucket
    // reduce (2^(c-1) buckets, G2 first).  Automatic: a group gets tables when its MSM has >= 2^20 points and the tables of
    // the groups chosen so far fit in a third of the free device memory (smallest first: Z, B, A+K); the rest stays for the
    // contexts' workspaces.  ctx->fixed_knob (mi_debug_set_prove_fixed_base): 0 = automatic, 1 = never, 17..22 = forced.
    {
        size_t free_b = 0, total_b = 0;
        if (rc == MI_OK && hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = 0;
        size_t budget = free_b / 3;
        auto nwin_of = [](u32 c) { return (size_t)((256 + c - 1) / c); };
        auto choose = [&](u32 knob, u32 c_auto, size_t n_max, size_t bytes_per_point) -> u32 {
            if (knob == 1) return 0;
            if (knob >= 17 && knob <= 22) return knob;
            const size_t need = nwin_of(c_auto) * bytes_per_point;
            if (n_max < ((size_t)1 << 20) || need > budget) return 0;
            budget -= need;
            return c_auto;
        };
        pk->c_z = choose(ctx->fixed_knob[2], 20, pk->n_z_msm, pk->n_z_msm * sizeof(G1Aff));
        pk->c_b = choose(ctx->fixed_knob[1], 17, pk->n_b, pk->n_b * (sizeof(G1Aff) + sizeof(G2Aff)));
        pk->c_ak = choose(ctx->fixed_knob[0], 19, pk->nb_wires, pk->nb_wires * 2 * sizeof(G1Aff));
        auto pre = [&](void **dst, const void *base, size_t n, int curve, u32 c) -> int32_t {
            const size_t bytes = nwin_of(c) * n * (curve == 1 ? sizeof(G1Aff) : sizeof(G2Aff));
            MI_CHECK_HIP(ctx, hipMalloc(dst, bytes ? bytes : 64));
            return mi_msm_precompute(ctx, curve, base, *dst, n, c);
        };
        // a group whose tables cannot be allocated after all (the budget is an estimate) falls back to the generic path
        auto group = [&](u32 &c, void **t0, const void *b0, int curve0, void **t1, const void *b1, int curve1, size_t n) -> int32_t {
            if (!c) return MI_OK;
            int32_t r = pre(t0, b0, n, curve0, c);
            if (r == MI_OK && t1) r = pre(t1, b1, n, curve1, c);
            if (r == MI_ENOMEM) {
                (void)hipGetLastError();
                if (*t0) { (void)hipFree(*t0); *t0 = nullptr; }
                if (t1 && *t1) { (void)hipFree(*t1); *t1 = nullptr; }
                c = 0;
                r = MI_OK;
            }
            return r;
        };
        if (rc == MI_OK) rc = group(pk->c_z, &pk->pre_z, pk->g1_z, 1, nullptr, nullptr, 0, pk->n_z_msm);
        if (rc == MI_OK) rc = group(pk->c_b, &pk->pre_b1, pk->g1_b, 1, &pk->pre_b2, pk->g2_b, 2, pk->n_b);
        if (rc == MI_OK) rc = group(pk->c_ak, &pk->pre_a, pk->a_full, 1, &pk->pre_k, pk->k_full, 1, pk->nb_wires);
    }
    if (rc == MI_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) { mi_set_err(ctx, "pk upload sync failed"); rc = MI_EHIP; }
    if (rc != MI_OK) { if (device_points && !adopt) pk->owns_points = false; mi_pk_free(ctx, pk); return rc; }
    // the compact A and K copies are not needed any more when the library owns them; nor are plain bases that have tables
    if (pk->owns_points) { (void)hipFree(pk->g1_a); (void)hipFree(pk->g1_k); pk->g1_a = pk->g1_k = nullptr; }
    if (pk->c_ak) { (void)hipFree(pk->a_full); (void)hipFree(pk->k_full); pk->a_full = pk->k_full = nullptr; }
    if (pk->owns_points && pk->c_b) { (void)hipFree(pk->g1_b); (void)hipFree(pk->g2_b); pk->g1_b = pk->g2_b = nullptr; }
    if (pk->owns_points && pk->c_z) { (void)hipFree(pk->g1_z); pk->g1_z = nullptr; }
    // the G1 arrays of the level-1 accumulation go into the R' packed form, in place where the key owns them
    if (mi_msm_limb29_enabled(ctx)) {
        const MsmCurveOps &g1 = mi_msm_ops(1);
        auto nwin_of = [](u32 c) { return (size_t)((256 + c - 1) / c); };
        auto in_place = [&](void *arr, size_t n) { if (arr && n) g1.to_rprime(ctx->stream, arr, arr, n); };
        auto own_or_copy = [&](void **arr, void **copy, size_t n) -> int32_t {
            if (!*arr || !n) return MI_OK;
            if (pk->owns_points) { g1.to_rprime(ctx->stream, *arr, *arr, n); return MI_OK; }
            MI_CHECK_HIP(ctx, hipMalloc(copy, n * sizeof(G1Aff)));
            g1.to_rprime(ctx->stream, *copy, *arr, n);
            *arr = *copy;
            return MI_OK;
        };
        int32_t r = MI_OK;
        if (pk->c_ak) { in_place(pk->pre_a, nwin_of(pk->c_ak) * pk->nb_wires); in_place(pk->pre_k, nwin_of(pk->c_ak) * pk->nb_wires); }
        else { in_place(pk->a_full, pk->nb_wires); in_place(pk->k_full, pk->nb_wires); }
        if (pk->c_b) in_place(pk->pre_b1, nwin_of(pk->c_b) * pk->n_b);
        else r = own_or_copy(&pk->g1_b, &pk->b1_copy, pk->n_b);
        {   // pk.G2.B the same way with the G2 conversion
            const MsmCurveOps &g2 = mi_msm_ops(2);
            if (pk->c_b) { if (pk->pre_b2 && pk->n_b) g2.to_rprime(ctx->stream, pk->pre_b2, pk->pre_b2, nwin_of(pk->c_b) * pk->n_b); }
            else if (r == MI_OK && pk->g2_b && pk->n_b) {
                if (pk->owns_points) g2.to_rprime(ctx->stream, pk->g2_b, pk->g2_b, pk->n_b);
                else if (hipMalloc(&pk->b2_copy, pk->n_b * sizeof(G2Aff)) != hipSuccess) { (void)hipGetLastError(); mi_set_err(ctx, "pk: no room for the converted copy of pk.G2.B"); r = MI_ENOMEM; }
                else { g2.to_rprime(ctx->stream, pk->b2_copy, pk->g2_b, pk->n_b); pk->g2_b = pk->b2_copy; }
            }
        }
        if (pk->c_z) in_place(pk->pre_z, nwin_of(pk->c_z) * pk->n_z_msm);
        else if (r == MI_OK) r = own_or_copy(&pk->g1_z, &pk->z_copy, pk->n_z_msm);
        if (r == MI_OK && (hipGetLastError() != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess)) { mi_set_err(ctx, "pk: conversion of the G1 arrays failed"); r = MI_EHIP; }
        if (r != MI_OK) { if (device_points && !adopt) pk->owns_points = false; mi_pk_free(ctx, pk); return r; }
        pk->rprime = true;
    }
    *out = pk;
    return MI_OK;
}

// k * p on the host, k canonical 8 x u32
template <class F>
static XYZZ<F> host_scalar_mul(const Affine<F> &p, const Fr &k_canon) { return xyzz_mul_256(XYZZ<F>::from_affine(p), k_canon.l); }

// ---------------------------------------------------------------- BSB22 Pedersen key
struct mi_pedersen_pk {
    void *basis = nullptr, *basis_exp_sigma = nullptr;
    size_t n = 0;
    // fixed-base window tables of both arrays (as for pk.G1.*: ONE bucket set for all windows, the level-1 additions in 29-bit limbs over
    // the R' form) when the key has >= 2^15 points and they fit: a commitment over 2^18 WHIR-mix values is then 15 windows into 2^16 buckets
    // instead of 19 windows x 2^13 -- the bucket reduce of the generic plan was most of what a Pedersen MSM cost a busy GPU (r4: 3.5 % of
    // the proofs/s at 2^18 committed wires, 8.6 % at 2^20)
    void *pre_basis = nullptr, *pre_sigma = nullptr;
    u32 c_tab = 0;
};
static constexpr u32 PEDERSEN_TABLE_C = 17;
// best effort: without room (or below 2^15 points) the key stays on the generic path
static void pedersen_build_tables(mi_ctx *ctx, mi_pedersen_pk *pk) {
    if (pk->n < ((size_t)1 << 15) || !mi_msm_limb29_enabled(ctx)) return;
    const u32 c = PEDERSEN_TABLE_C;
    const size_t nwin = (256 + c - 1) / c, bytes = nwin * pk->n * sizeof(G1Aff);
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || 2 * bytes > free_b / 4) { (void)hipGetLastError(); return; }
    void *t[2] = {nullptr, nullptr};
    const void *src[2] = {pk->basis, pk->basis_exp_sigma};
    bool ok = true;
    for (int k = 0; k < 2 && ok; k++) {
        ok = hipMalloc(&t[k], bytes) == hipSuccess && mi_msm_precompute(ctx, 1, src[k], t[k], pk->n, c) == MI_OK;
        if (ok) mi_msm_ops(1).to_rprime(ctx->stream, t[k], t[k], nwin * pk->n);
    }
    if (ok) ok = hipGetLastError() == hipSuccess && hipStreamSynchronize(ctx->stream) == hipSuccess;
    if (!ok) { (void)hipGetLastError(); for (void *q : t) if (q) (void)hipFree(q); return; }
    pk->pre_basis = t[0]; pk->pre_sigma = t[1]; pk->c_tab = c;
}
// One Pedersen MSM on slot 5 of ctx (its own stream: it runs beside the five MSMs of a proof), in two halves so that a prover-pool job
// can enqueue its proof's ProveKnowledge MSM, prove, and collect (pool.hip); host values in, affine point out.
static int32_t pedersen_enqueue(mi_ctx *ctx, const void *bases_dev, const void *tables_dev, u32 c_tab, size_t key_n, const mi_fr *values, size_t n) {
    if (!ctx || (!values && n)) return MI_EINVAL;
    if (n > key_n) MI_FAIL(ctx, MI_EINVAL, "pedersen: more values than basis points");   // gnark: "must have as many values as basis elements"
    MI_TRY(mi_reserve(ctx, ctx->ws[19], n * sizeof(mi_fr) + 64));
    if (n) MI_CHECK_HIP(ctx, hipMemcpyAsync(ctx->ws[19].p, values, n * sizeof(mi_fr), hipMemcpyHostToDevice, ctx->msm[5].stream));
    // the tables are laid out [window][key_n]: they serve exactly the full-length call (gnark's: as many values as basis elements)
    if (tables_dev && n == key_n) return mi_msm_enqueue(ctx, 5, -1, 1, tables_dev, ctx->ws[19].p, n, MI_MSM_PTS_RPRIME, nullptr, false, c_tab);
    return mi_msm_enqueue(ctx, 5, -1, 1, bases_dev, ctx->ws[19].p, n, 0, nullptr, false);
}
static int32_t pedersen_collect(mi_ctx *ctx, mi_g1_affine *out) {
    G1X r;
    MI_TRY(mi_msm_finish(ctx, 5, 1, &r));
    G1Aff a = xyzz_to_affine(r);
    if (out) std::memcpy(out, &a, sizeof(a));
    return MI_OK;
}
static int32_t pedersen_msm(mi_ctx *ctx, const mi_pedersen_pk *pk, bool sigma, const mi_fr *values, size_t n, mi_g1_affine *out) {
    if (!out) return MI_EINVAL;
    MI_TRY(pedersen_enqueue(ctx, sigma ? pk->basis_exp_sigma : pk->basis, sigma ? pk->pre_sigma : pk->pre_basis, pk->c_tab, pk->n, values, n));
    return pedersen_collect(ctx, out);
}
int32_t mi_pedersen_pok_enqueue(mi_ctx *ctx, mi_pedersen_pk *pk, const mi_fr *values, size_t n) {
    if (!pk) return MI_EINVAL;
    return pedersen_enqueue(ctx, pk->basis_exp_sigma, pk->pre_sigma, pk->c_tab, pk->n, values, n);
}
int32_t mi_pedersen_pok_collect(mi_ctx *ctx, mi_g1_affine *pok) { return pedersen_collect(ctx, pok); }

extern "C" {
int32_t mi_pedersen_pk_load(mi_ctx *ctx, const mi_g1_affine *basis, const mi_g1_affine *basis_exp_sigma, size_t n, mi_pedersen_pk **out) {
    if (!ctx || !out || ((!basis || !basis_exp_sigma) && n)) return MI_EINVAL;
    *out = nullptr;
    mi_pedersen_pk *pk = new (std::nothrow) mi_pedersen_pk();
    if (!pk) return MI_ENOMEM;
    pk->n = n;
    int32_t rc = upload(ctx, &pk->basis, basis, n * 64);
    if (rc == MI_OK) rc = upload(ctx, &pk->basis_exp_sigma, basis_exp_sigma, n * 64);
    if (rc == MI_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) { mi_set_err(ctx, "pedersen key upload failed"); rc = MI_EHIP; }
    if (rc != MI_OK) { mi_pedersen_pk_free(ctx, pk); return rc; }
    pedersen_build_tables(ctx, pk);
    *out = pk;
    return MI_OK;
}
int32_t mi_pedersen_pk_adopt(mi_ctx *ctx, void *basis_dev, void *basis_exp_sigma_dev, size_t n, mi_pedersen_pk **out) {
    if (!ctx || !out) return MI_EINVAL;
    mi_pedersen_pk *pk = new (std::nothrow) mi_pedersen_pk();
    if (!pk) return MI_ENOMEM;
    pk->basis = basis_dev; pk->basis_exp_sigma = basis_exp_sigma_dev; pk->n = n;
    pedersen_build_tables(ctx, pk);
    *out = pk;
    return MI_OK;
}
int32_t mi_pedersen_pk_free(mi_ctx *ctx, mi_pedersen_pk *pk) {
    if (!ctx || !pk) return MI_EINVAL;
    (void)hipStreamSynchronize(ctx->stream);
    for (void *q : {pk->basis, pk->basis_exp_sigma, pk->pre_basis, pk->pre_sigma}) if (q) (void)hipFree(q);
    delete pk;
    return MI_OK;
}
int32_t mi_pedersen_commit(mi_ctx *ctx, mi_pedersen_pk *pk, const mi_fr *values, size_t n, mi_g1_affine *commitment) {
    if (!pk) return MI_EINVAL;
    return pedersen_msm(ctx, pk, false, values, n, commitment);
}
int32_t mi_pedersen_prove_knowledge(mi_ctx *ctx, mi_pedersen_pk *pk, const mi_fr *values, size_t n, mi_g1_affine *pok) {
    if (!pk) return MI_EINVAL;
    return pedersen_msm(ctx, pk, true, values, n, pok);
}
// sum_i challenge^i * points[i] on the host (a handful of points: one per commitment)
int32_t mi_pedersen_fold(const mi_g1_affine *points, size_t n, const mi_fr *challenge, mi_g1_affine *out) {
    if ((!points && n) || !challenge || !out) return MI_EINVAL;
    Fr ch, pw = Fr::one();
    std::memcpy(&ch, challenge, 32);
    G1X acc = G1X::inf();
    for (size_t i = 0; i < n; i++) {
        G1Aff p;
        std::memcpy(&p, &points[i], sizeof(p));
        G1X t = i ? host_scalar_mul<Fp>(p, fe_from_mont(pw)) : G1X::from_affine(p);   // challenge^0 = 1: no 256-bit ladder for the first (usually only) point
        xyzz_add(acc, t);
        pw = pw * ch;
    }
    if (n == 1) { std::memcpy(out, &points[0], sizeof(*out)); return MI_OK; }   // one commitment: the fold is the point itself, bit for bit
    G1Aff a = xyzz_to_affine(acc);
    std::memcpy(out, &a, sizeof(a));
    return MI_OK;
}
int32_t mi_get_mem_ledger(mi_ctx *ctx, const mi_pk *pk, mi_mem_ledger *out) {
    if (!ctx || !out) return MI_EINVAL;
    std::memset(out, 0, sizeof(*out));
    auto nwin_of = [](u32 c) { return (size_t)((256 + c - 1) / c); };
    if (pk) {
        const size_t g1 = sizeof(G1Aff), g2 = sizeof(G2Aff);
        if (pk->owns_points) out->key_bases += (pk->g1_a ? pk->n_a * g1 : 0) + (pk->g1_k ? pk->n_k * g1 : 0) + (pk->g1_b ? pk->n_b * g1 : 0) + (pk->g2_b ? pk->n_b * g2 : 0) + (pk->g1_z ? pk->n_z * g1 : 0);
        out->key_bases += (pk->a_full ? pk->nb_wires * g1 : 0) + (pk->k_full ? pk->nb_wires * g1 : 0);
        out->key_bases += (pk->b1_copy ? pk->n_b * g1 : 0) + (pk->b2_copy ? pk->n_b * g2 : 0) + (pk->z_copy ? pk->n_z_msm * g1 : 0);
        if (pk->c_ak) out->key_tables += 2 * nwin_of(pk->c_ak) * pk->nb_wires * g1;
        if (pk->c_b) out->key_tables += nwin_of(pk->c_b) * pk->n_b * (g1 + g2);
        if (pk->c_z) out->key_tables += nwin_of(pk->c_z) * pk->n_z_msm * g1;
        out->key_indices = (pk->n_a + pk->n_b + pk->n_k) * 4;
    }
    out->ctx_ntt_tables = mi_ntt_table_bytes(ctx);
    for (int i = 0; i < 24; i++) {
        const size_t cap = ctx->ws[i].cap;
        if (i == 0 || i == 1 || i == 14) out->ctx_ntt_vectors += cap; else out->ctx_other += cap;
    }
    for (const auto &sl : ctx->msm) for (const auto &b : sl.buf) out->ctx_msm += b.cap;
    return MI_OK;
}
int32_t mi_pk_table_plan(const mi_pk *pk, uint32_t c_out[3]) {
    if (!pk || !c_out) return MI_EINVAL;
    c_out[0] = pk->c_ak; c_out[1] = pk->c_b; c_out[2] = pk->c_z;
    return MI_OK;
}
int32_t mi_pk_load(mi_ctx *ctx, const mi_pk_desc *d, mi_pk **out) {
    if (d && (!d->g1_a && d->n_g1_a)) return MI_EINVAL;
    return mi_pk_load_range(ctx, d, out, false, nullptr);
}
int32_t mi_pk_load_dev(mi_ctx *ctx, const mi_pk_desc *d, mi_pk **out) { return mi_pk_load_range(ctx, d, out, true, nullptr); }
int32_t mi_pk_free(mi_ctx *ctx, mi_pk *pk) {
    if (!ctx || !pk) return MI_EINVAL;
    (void)hipStreamSynchronize(ctx->stream);
    if (pk->owns_points) for (void *p : {pk->g1_a, pk->g1_b, pk->g1_k, pk->g1_z, pk->g2_b}) if (p) (void)hipFree(p);
    for (void *p : {(void *)pk->idx_a, (void *)pk->idx_b, (void *)pk->idx_k, (void *)pk->a_full, (void *)pk->k_full}) if (p) (void)hipFree(p);
    for (void *p : {pk->pre_a, pk->pre_k, pk->pre_b1, pk->pre_b2, pk->pre_z, pk->b1_copy, pk->z_copy, pk->b2_copy}) if (p) (void)hipFree(p);
    delete pk;
    return MI_OK;
}

}  // extern "C"

// ---------------------------------------------------------------- the pieces of a proof (shared with group.hip)
// step 5 + the wire MSMs: they depend on W only (ev_w = "W is on the device").  Two independent groups, each with ONE sort:
// B1 + B2 (slots 1, 2) and A + K (slots 0, 3).  Enqueueing a group waits once, on the host, for the count pass of its sort (msm.hip,
// MI_MSM_EXACT_SIZE), so a proof enqueues the two groups from two helper threads while its own thread enqueues computeH and the Z MSM
// (prove_common): measured on one proof alone, the A + K sort used to start 10 ms into the proof because the host was still busy
// enqueueing Z's and B's ~130 launches, and the memory-bound sorts then ran beside the bucket accumulations instead of beside the NTT.
int32_t mi_prove_enqueue_b_msms(mi_ctx *ctx, mi_pk *pk, const mi_fr *W, hipEvent_t ev_w, bool defer, const std::function<hipEvent_t()> *accum_gate) {
    (void)hipSetDevice(ctx->dev);   // the current device is per host thread
    struct Gates { mi_ctx *c; const std::function<hipEvent_t()> *g; void arm(int slot) const { c->msm[slot].accum_gate = g; } } gates{ctx, accum_gate};
    // wire values are skewed (45 % of them 0 or 1): their sorts are sized by the counted entries, not by windows * n (msm.hip)
    const uint32_t df = (defer ? MI_MSM_DEFER_REDUCE : 0) | MI_MSM_EXACT_SIZE;
    const uint32_t rp = pk->rprime ? MI_MSM_PTS_RPRIME : 0;
    // wireValuesB by the static gather indices, on its MSM's stream; B2 (G2) shares B1's sort (same scalars)
    MI_TRY(mi_reserve(ctx, ctx->ws[17], (pk->n_b + 1) * sizeof(Fr)));
    hipStream_t st = ctx->msm[1].stream;
    if (ev_w) MI_CHECK_HIP(ctx, hipStreamWaitEvent(st, ev_w, 0));
    if (pk->n_b) hipLaunchKernelGGL(k_gather_fr, dim3((unsigned)((pk->n_b + 255) / 256)), dim3(256), 0, st, (Fr *)ctx->ws[17].p, (const Fr *)W, pk->idx_b, pk->n_b);
    MI_CHECK_HIP(ctx, hipGetLastError());
    if (pk->pre_b1) {
        gates.arm(1);
        // (B1 and K are not timed: their chains interleave on one stream, so the event pair around one's level-1 launch may bracket kernels
        //  of the other; mi_stats.g1_accum_* then cover A and Z, the two launches with a stream of their own)
        MI_TRY(mi_msm_enqueue(ctx, 1, -1, 1, pk->pre_b1, ctx->ws[17].p, pk->n_b, df | rp, nullptr, false, pk->c_b));
        gates.arm(2);
        return mi_msm_enqueue(ctx, 2, 1, 2, pk->pre_b2, nullptr, pk->n_b, df | rp, nullptr, false, pk->c_b);
    }
    gates.arm(1);
    MI_TRY(mi_msm_enqueue(ctx, 1, -1, 1, pk->g1_b, ctx->ws[17].p, pk->n_b, df | rp, nullptr, false, 0, 0, pk->gen_c_b));
    gates.arm(2);
    return mi_msm_enqueue(ctx, 2, 1, 2, pk->g2_b, nullptr, pk->n_b, df | rp, nullptr, false);
}
int32_t mi_prove_enqueue_ak_msms(mi_ctx *ctx, mi_pk *pk, const mi_fr *W, hipEvent_t ev_w, bool defer, const std::function<hipEvent_t()> *accum_gate) {
    (void)hipSetDevice(ctx->dev);
    struct Gates { mi_ctx *c; const std::function<hipEvent_t()> *g; void arm(int slot) const { c->msm[slot].accum_gate = g; } } gates{ctx, accum_gate};
    const uint32_t df = (defer ? MI_MSM_DEFER_REDUCE : 0) | MI_MSM_EXACT_SIZE;
    const uint32_t rp = pk->rprime ? MI_MSM_PTS_RPRIME : 0;
    // A and K are both multiplied by W itself: one sort of all wires (slot 0) serves both, against the per-wire expanded
    // point arrays (a wire without a point reads (0,0) = infinity and is skipped); no gather, one sort less
    if (pk->pre_a) {
        gates.arm(0);
        MI_TRY(mi_msm_enqueue(ctx, 0, -1, 1, pk->pre_a, W, pk->nb_wires, df | rp, ev_w, true, pk->c_ak, pk->n_a));
        gates.arm(3);
        return mi_msm_enqueue(ctx, 3, 0, 1, pk->pre_k, nullptr, pk->nb_wires, df | rp, nullptr, false, pk->c_ak, pk->n_k);   // (not timed: K and B1 share a stream, msm.hip)
    }
    gates.arm(0);
    MI_TRY(mi_msm_enqueue(ctx, 0, -1, 1, pk->a_full, W, pk->nb_wires, df | rp, ev_w, true, 0, pk->n_a, pk->gen_c_ak));
    gates.arm(3);
    return mi_msm_enqueue(ctx, 3, 0, 1, pk->k_full, nullptr, pk->nb_wires, df | rp, nullptr, false, 0, pk->n_k);
}
// both groups from two helper threads; returns when everything is enqueued
int32_t mi_prove_enqueue_wire_msms(mi_ctx *ctx, mi_pk *pk, const mi_fr *W, hipEvent_t ev_w, bool defer) {
    std::future<int32_t> fb;
    int32_t rb = MI_OK;
    try {
        fb = std::async(std::launch::async, [=]() -> int32_t { try { return mi_prove_enqueue_b_msms(ctx, pk, W, ev_w, defer, nullptr); } catch (...) { return MI_ENOMEM; } });
    } catch (...) {
        rb = mi_prove_enqueue_b_msms(ctx, pk, W, ev_w, defer, nullptr);
    }
    const int32_t ra = mi_prove_enqueue_ak_msms(ctx, pk, W, ev_w, defer, nullptr);
    if (fb.valid()) rb = fb.get();
    return ra != MI_OK ? ra : rb;
}
// the Z MSM over this key's h coefficients against the bit-reversed pk.G1.Z (ev_h = "h is ready")
int32_t mi_prove_enqueue_z_msm(mi_ctx *ctx, mi_pk *pk, const mi_fr *h, hipEvent_t ev_h, bool defer) {
    const uint32_t df = (defer ? MI_MSM_DEFER_REDUCE : 0) | (pk->rprime ? MI_MSM_PTS_RPRIME : 0);
    if (pk->pre_z) return mi_msm_enqueue(ctx, 4, -1, 1, pk->pre_z, h, pk->n_z_msm, df, ev_h, true, pk->c_z);
    return mi_msm_enqueue(ctx, 4, -1, 1, pk->g1_z, h, pk->n_z_msm, df, ev_h, true, 0, 0, pk->gen_c_z);
}

// a host helper thread for one independent scalar multiplication; without a thread to be had the work runs here (no exception may
// cross the C-ABI: std::async throws std::system_error when the process is out of threads)
template <class Fn>
static auto async_or_inline(Fn fn) -> std::future<decltype(fn())> {
    try {
        return std::async(std::launch::async, fn);
    } catch (...) {
        std::promise<decltype(fn())> p;
        p.set_value(fn());
        return p.get_future();
    }
}
void ProofAssembler::start(const mi_pk *pk_, const mi_fr *r_m, const mi_fr *s_m) {
    pk = pk_;
    Fr r, s;
    std::memcpy(&r, r_m, 32); std::memcpy(&s, s_m, 32);
    rc = fe_from_mont(r); sc = fe_from_mont(s); krc = fe_from_mont(fe_neg(r * s));
    // (independent 256-bit scalar multiplications: one host thread each; for small circuits they are the longest chain)
    f_r = async_or_inline([this] { return host_scalar_mul<Fp>(pk->delta1, rc); });
    f_s = async_or_inline([this] { return host_scalar_mul<Fp>(pk->delta1, sc); });
    f_kr = async_or_inline([this] { return host_scalar_mul<Fp>(pk->delta1, krc); });
    s_delta2 = host_scalar_mul<Fp2>(pk->delta2, sc);
    r_delta = f_r.get(); s_delta = f_s.get(); kr_delta = f_kr.get();
}
void ProofAssembler::have_a_b1(const G1X &msm_a, const G1X &msm_b1) {
    G1X ar = msm_a;
    xyzz_madd(ar, pk->alpha1, false);
    xyzz_add(ar, r_delta);
    G1X bs1 = msm_b1;
    xyzz_madd(bs1, pk->beta1, false);
    xyzz_add(bs1, s_delta);
    ar_aff = xyzz_to_affine(ar); bs1_aff = xyzz_to_affine(bs1);
    f_sar = async_or_inline([this] { return host_scalar_mul<Fp>(ar_aff, sc); });   // overlaps the remaining MSMs
    r_bs1 = host_scalar_mul<Fp>(bs1_aff, rc);
}
void ProofAssembler::finish(const G1X &msm_k, const G2X &msm_b2, const G1X &msm_z, mi_proof_out *out) {
    G1X s_ar = f_sar.get();
    G1X krs = msm_k;
    xyzz_add(krs, msm_z);
    xyzz_add(krs, kr_delta);
    xyzz_add(krs, s_ar);
    xyzz_add(krs, r_bs1);
    G2X bs = msm_b2;
    xyzz_madd(bs, pk->beta2, false);
    xyzz_add(bs, s_delta2);
    G1Aff krs_aff = xyzz_to_affine(krs);
    G2Aff bs_aff = xyzz_to_affine(bs);
    std::memcpy(&out->ar, &ar_aff, 64); std::memcpy(&out->bs, &bs_aff, 128); std::memcpy(&out->krs, &krs_aff, 64);
}

// Host-pointer inputs of mi_groth16_prove (null for the device-pointer entry point).
struct HostInputs { const mi_fr *W, *a, *b, *c; };
// Device inputs that are still ARRIVING (the prover pool's upload stage, pool.hip): W is resident when the call is made; abc() blocks
// the host until a, b, c are resident too and returns true (false: their upload failed).  Host-side waits on purpose: events recorded
// on the pool's copy stream between its pageable copies slowed those copies down (round 3: uploads of 50-70 ms instead of 19).
struct AbcGate { const std::function<bool(int)> *abc; /* abc(k): blocks until k of a, b, c are resident */ bool abc_arrived; /* they all were when the job was picked up */ };

// W, a, b, c: device buffers (for host inputs: staging areas the uploads below fill).
static int32_t prove_common(mi_ctx *ctx, mi_pk *pk, const mi_fr *W, size_t n_wires, const mi_fr *a, const mi_fr *b, const mi_fr *c,
                            size_t n_constraints, const mi_fr *r_m, const mi_fr *s_m, mi_proof_out *out, mi_stats *stats, const HostInputs *host,
                            const AbcGate *gate = nullptr) {
    if (!ctx || !pk || !r_m || !s_m || !out) return MI_EINVAL;
    const MiRange range_all("mi.prove");
    // null W / a / b only where the matching count is 0 (the header's rule, as the pool's submit applies it); c == null: c = a o b on the device
    if ((!W && n_wires) || ((!a || !b) && n_constraints)) MI_FAIL(ctx, MI_EINVAL, "prove: null W, a or b with a non-zero count");
    if (host && ((!host->W && n_wires) || ((!host->a || !host->b) && n_constraints))) MI_FAIL(ctx, MI_EINVAL, "prove: null host W, a or b with a non-zero count");
    const bool derive_c = host ? !host->c : !c;
    const size_t N = (size_t)1 << pk->log_n;
    if (pk->wire_lo || pk->n_z_msm != N - 1) MI_FAIL(ctx, MI_EINVAL, "prove: this key is one part of a sharded key (use mi_groth16_prove_sharded)");
    if (n_wires != pk->nb_wires || n_constraints > N) MI_FAIL(ctx, MI_EINVAL, "prove: witness size does not match the proving key");
    std::memset(&ctx->stats, 0, sizeof(ctx->stats));
    const auto t_begin = std::chrono::steady_clock::now();
    hipEvent_t *ev = ctx->ev;
    // Stream plan: computeH on the caller's stream; MSM A, B1, B2, K, Z on slots 0..4 (own streams).  A, B and
    // K depend only on W and start at once; B2 (G2) reuses B1's sort; Z waits for h.  The latency-bound tails
    // (levels >= 2, bucket reduce, scans) of one MSM overlap the throughput-bound accumulation of the others.
    MI_TRY(mi_reserve(ctx, ctx->ws[14], N * sizeof(Fr)));
    Fr *h = (Fr *)ctx->ws[14].p;
    // step 4: h = computeH(a, b, c)  (bit-reversed, like gnark leaves it), then the Z MSM over h[:N-1] against the bit-reversed pk.G1.Z
    // "computeH is enqueued and its end event recorded": what the wire MSMs' accumulations wait for (hold, below).  Always fulfilled
    // before the helper threads are joined -- with a null event on a failure path, which releases them without a wait.
    std::promise<hipEvent_t> h_recorded;
    bool h_promised = false;
    const std::shared_future<hipEvent_t> h_fut = h_recorded.get_future().share();
    const std::function<hipEvent_t()> h_gate = [h_fut] { return h_fut.get(); };
    // the Z MSM's digit count rides in computeH's last launch (mi_ctx::zhook): armed for exactly that launch
    auto arm_z_count = [&]() -> int32_t { return pk->pre_z ? mi_msm_z_count_arm(ctx, MI_ZHOOK_SLOT, pk->n_z_msm, pk->c_z) : MI_OK; };
    auto last_part_of_h = [&]() -> int32_t {
        MI_TRY(arm_z_count());
        const int32_t rc_h = mi_compute_h_part(ctx, pk->log_n, 3, nullptr, n_constraints, (mi_fr *)h);
        ctx->zhook.armed = false;
        return rc_h;
    };
    auto enqueue_h_and_z = [&]() -> int32_t {
        MI_TRY(arm_z_count());
        const int32_t rc_h = mi_compute_h_dev_impl(ctx, pk->log_n, a, b, derive_c ? nullptr : c, n_constraints, (mi_fr *)h);
        ctx->zhook.armed = false;
        MI_TRY(rc_h);
        MI_CHECK_HIP(ctx, hipEventRecord(ev[3], ctx->stream));
        h_recorded.set_value(ev[3]); h_promised = true;
        return mi_prove_enqueue_z_msm(ctx, pk, (const mi_fr *)h, ev[3]);
    };
    // The two wire-MSM groups are enqueued by two helper threads (each waits once for its sort's count pass) while this thread
    // enqueues computeH and the Z MSM -- or is held by the uploads of a, b, c: all sorts start at the head of the proof, beside the NTT.
    // Whatever fails, the helpers are joined and every slot is collected before the error is returned (nothing of this proof may stay
    // queued on the slots).
    // hold: the wire MSMs SORT at once (memory-bound, beside the NTT) but their bucket accumulations wait for computeH.  Measured on
    // one proof alone: let loose, the accumulations' resident waves starved the NTT passes (stream priorities order dispatch, they do
    // not preempt), computeH ended at 18 instead of 9 ms, the Z MSM's sort -- which needs h -- could no longer hide beside the other
    // accumulations and the GPU idled for 3 ms before the Z accumulation.  Not for host inputs whose a, b, c are still on the PCIe
    // bus: there the accumulations are what fills the GPU meanwhile.
    const bool hold = ctx->hold_accum && !host && (!gate || gate->abc_arrived);
    std::future<int32_t> f_b, f_ak;
    int32_t rc_inline = MI_OK;
    auto start_wires = [&](hipEvent_t ev_w) {   // ev_w: "W is resident" on some stream, or null when the host already knows it is
        const std::function<hipEvent_t()> *g = hold ? &h_gate : nullptr;
        try {
            f_b = std::async(std::launch::async, [=]() -> int32_t { try { return mi_prove_enqueue_b_msms(ctx, pk, W, ev_w, false, g); } catch (...) { return MI_ENOMEM; } });
        } catch (...) {   // no thread to be had: this thread does it (without the hold: it would wait for itself); no exception crosses the C-ABI
            rc_inline = mi_prove_enqueue_b_msms(ctx, pk, W, ev_w, false, nullptr);
        }
        try {
            f_ak = std::async(std::launch::async, [=]() -> int32_t { try { return mi_prove_enqueue_ak_msms(ctx, pk, W, ev_w, false, g); } catch (...) { return MI_ENOMEM; } });
        } catch (...) {
            const int32_t r = mi_prove_enqueue_ak_msms(ctx, pk, W, ev_w, false, nullptr);
            if (rc_inline == MI_OK) rc_inline = r;
        }
    };
    auto main_part = [&]() -> int32_t {
        if (gate) {
            // inputs on their way into HBM (pool upload stage): the wire MSMs start as soon as W is there; a, b, c (3/4 of the bytes)
            // finish arriving behind them, and computeH waits for exactly that
            start_wires(nullptr);   // W is resident (the pool synchronised its copy stream before handing the job over)
            MI_CHECK_HIP(ctx, hipEventRecord(ev[11], ctx->stream));
            // computeH one vector at a time, as the vectors arrive (the direct host-pointer path below does the same): a's transforms
            // run while b is on the bus; h is ready ~2.5 ms after the last vector instead of a whole computeH later
            const auto need = [&](int k) -> int32_t { if (!(*gate->abc)(k)) MI_FAIL(ctx, MI_EHIP, "prove: the upload of a, b, c failed"); return MI_OK; };
            MI_TRY(need(1));
            MI_TRY(mi_compute_h_part(ctx, pk->log_n, 0, a, n_constraints, (mi_fr *)h));
            MI_TRY(need(2));
            MI_TRY(mi_compute_h_part(ctx, pk->log_n, 1, b, n_constraints, (mi_fr *)h));
            if (derive_c) MI_TRY(mi_compute_h_part(ctx, pk->log_n, 2, a, n_constraints, (mi_fr *)h, b));
            else { MI_TRY(need(3)); MI_TRY(mi_compute_h_part(ctx, pk->log_n, 2, c, n_constraints, (mi_fr *)h)); }
            MI_TRY(last_part_of_h());
            MI_CHECK_HIP(ctx, hipEventRecord(ev[3], ctx->stream));
            h_recorded.set_value(ev[3]); h_promised = true;
            return mi_prove_enqueue_z_msm(ctx, pk, (const mi_fr *)h, ev[3]);
        }
        if (!host) {
            // inputs already in HBM
            MI_CHECK_HIP(ctx, hipEventRecord(ev[2], ctx->stream));
            start_wires(ev[2]);
            return enqueue_h_and_z();
        }
        // inputs in host memory (the cgo path): upload W, start the wire MSMs, and upload a, b, c WHILE they run; the
        // PCIe time of a, b, c (3/4 of the bytes) disappears behind the MSMs instead of preceding the whole proof.  The copies run on
        // the context's copy stream, which carries nothing else, and are ordered by synchronising it on this thread (a pageable copy
        // holds the calling thread anyway): events between them cost 2-3.5x on every later copy (the pool's lesson).
        // Order on the bus: W, a, b, c.  The wire MSMs start as soon as W is resident; computeH heads the proof's longest chain (h -> Z
        // MSM), so a's inverse and coset transforms start as soon as a is resident (mi_compute_h_part), b's behind b, and what needs
        // all three behind c: h is ready ~2.5 ms after c has crossed instead of a whole computeH later.  (a first: measured 2 ms worse
        // -- the wire MSMs' 14 ms of work then start 5 ms later and nothing else can fill the bus time.)
        const size_t wb = n_wires * sizeof(mi_fr), cb = n_constraints * sizeof(mi_fr);
        const auto t_up = std::chrono::steady_clock::now();
        hipStream_t cps = nullptr;
        MI_TRY(mi_copy_stream(ctx, &cps));
        auto upload = [&](const mi_fr *dst, const mi_fr *src, size_t bytes) -> int32_t {
            const MiRange range("mi.prove.upload");
            if (bytes) MI_CHECK_HIP(ctx, hipMemcpyAsync((void *)dst, src, bytes, hipMemcpyHostToDevice, cps));
            MI_CHECK_HIP(ctx, hipStreamSynchronize(cps));
            return MI_OK;
        };
        MI_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));   // the staging area may still be read by an earlier call's kernels
        MI_CHECK_HIP(ctx, hipEventRecord(ev[11], ctx->stream));
        MI_TRY(upload(W, host->W, wb));
        start_wires(nullptr);   // W is resident: nothing to wait for
        MI_TRY(upload(a, host->a, cb));
        MI_TRY(mi_compute_h_part(ctx, pk->log_n, 0, a, n_constraints, (mi_fr *)h));
        MI_TRY(upload(b, host->b, cb));
        MI_TRY(mi_compute_h_part(ctx, pk->log_n, 1, b, n_constraints, (mi_fr *)h));
        if (!derive_c) MI_TRY(upload(c, host->c, cb));   // (derive_c: a quarter of the proof's PCIe bytes never crosses)
        ctx->stats.h2d_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_up).count();
        if (derive_c) MI_TRY(mi_compute_h_part(ctx, pk->log_n, 2, a, n_constraints, (mi_fr *)h, b));
        else MI_TRY(mi_compute_h_part(ctx, pk->log_n, 2, c, n_constraints, (mi_fr *)h));
        MI_TRY(last_part_of_h());
        MI_CHECK_HIP(ctx, hipEventRecord(ev[3], ctx->stream));
        h_recorded.set_value(ev[3]); h_promised = true;
        return mi_prove_enqueue_z_msm(ctx, pk, (const mi_fr *)h, ev[3]);
    };
    {
        int32_t rc = main_part();
        if (rc == MI_OK) rc = rc_inline;
        if (!h_promised) h_recorded.set_value(nullptr);   // a failure before computeH's event: release the helpers
        if (f_b.valid()) { const int32_t r = f_b.get(); if (rc == MI_OK) rc = r; }
        if (f_ak.valid()) { const int32_t r = f_ak.get(); if (rc == MI_OK) rc = r; }
        // a digit count computeH's last launch produced for a Z sort that never came must not meet a later sort of the same shape
        // (the sort also compares the counted vector's address)
        if (rc != MI_OK) ctx->zhook.armed = ctx->zhook.done = false;
        if (rc != MI_OK) {
            const std::string keep = ctx->err;   // the collection below may overwrite it
            G1X t1; G2X t2;
            for (int sl : {0, 1, 3, 4}) (void)mi_msm_finish(ctx, sl, 1, &t1);
            (void)mi_msm_finish(ctx, 2, 2, &t2);
            (void)hipStreamSynchronize(ctx->stream);
            mi_set_err(ctx, keep);
            return rc;
        }
    }
    // step 6 while the GPU works: blinding multiples of delta on the host (O(1) points)
    ProofAssembler as;
    { const MiRange range("mi.prove.blinding"); as.start(pk, r_m, s_m); }
    // step 7: collect the five MSMs
    G1X msm_a, msm_b1, msm_k, msm_z;
    G2X msm_b2;
    MI_TRY(mi_msm_finish(ctx, 0, 1, &msm_a));
    MI_TRY(mi_msm_finish(ctx, 1, 1, &msm_b1));
    const auto t_asm0 = std::chrono::steady_clock::now();
    as.have_a_b1(msm_a, msm_b1);
    MI_TRY(mi_msm_finish(ctx, 3, 1, &msm_k));
    MI_TRY(mi_msm_finish(ctx, 2, 2, &msm_b2));
    MI_TRY(mi_msm_finish(ctx, 4, 1, &msm_z));
    MI_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const auto t_gpu_done = std::chrono::steady_clock::now();
    { const MiRange range("mi.prove.assemble"); as.finish(msm_k, msm_b2, msm_z, out); }
    const auto t_end = std::chrono::steady_clock::now();
    mi_stats &st = ctx->stats;
    auto ms = [](std::chrono::steady_clock::time_point x, std::chrono::steady_clock::time_point y) {
        return std::chrono::duration<float, std::milli>(y - x).count();
    };
    auto slot_ms = [&](int slot, float *dst) -> int32_t {
        MI_CHECK_HIP(ctx, hipEventElapsedTime(dst, ctx->msm[slot].ev[3], ctx->msm[slot].ev[4]));
        return MI_OK;
    };
    // per-phase spans overlap (five streams): they do not add up to total_ms
    MI_CHECK_HIP(ctx, hipEventElapsedTime(&st.compute_h_ms, (host || gate) ? ev[11] : ev[2], ev[3]));
    // (h2d_ms of the host-pointer path: wall clock of the uploads, set above -- they overlap the wire MSMs)
    if (pk->nb_wires) MI_TRY(slot_ms(0, &st.msm_a_ms));
    if (pk->n_b) { MI_TRY(slot_ms(1, &st.msm_b1_ms)); MI_TRY(slot_ms(2, &st.msm_b2_ms)); }
    if (pk->nb_wires) MI_TRY(slot_ms(3, &st.msm_k_ms));
    if (N > 1) MI_TRY(slot_ms(4, &st.msm_z_ms));
    st.assemble_ms = ms(t_gpu_done, t_end);     // host work left after the last MSM landed
    st.filter_ms = ms(t_asm0, t_gpu_done);      // host blinding work hidden under the GPU
    st.total_ms = ms(t_begin, t_end);
    if (stats) *stats = st;
    return MI_OK;
}

int32_t mi_groth16_prove_dev_gated(mi_ctx *ctx, mi_pk *pk, const mi_fr *W, size_t n_wires, const mi_fr *a, const mi_fr *b, const mi_fr *c, size_t n_constraints,
                                   const mi_fr *r, const mi_fr *s, mi_proof_out *out, mi_stats *stats, const std::function<bool(int)> &abc_ready, bool abc_arrived) {
    const AbcGate gate{&abc_ready, abc_arrived};
    return prove_common(ctx, pk, W, n_wires, a, b, c, n_constraints, r, s, out, stats, nullptr, &gate);
}

extern "C" {
int32_t mi_groth16_prove_dev(mi_ctx *ctx, mi_pk *pk, const mi_fr *W, size_t n_wires, const mi_fr *a, const mi_fr *b, const mi_fr *c,
                             size_t n_constraints, const mi_fr *r, const mi_fr *s, mi_proof_out *out, mi_stats *stats) {
    return prove_common(ctx, pk, W, n_wires, a, b, c, n_constraints, r, s, out, stats, nullptr);
}
int32_t mi_groth16_prove(mi_ctx *ctx, mi_pk *pk, const mi_fr *W, size_t n_wires, const mi_fr *a, const mi_fr *b, const mi_fr *c,
                         size_t n_constraints, const mi_fr *r, const mi_fr *s, mi_proof_out *out, mi_stats *stats) {
    if (!ctx || !pk || !r || !s || !out) return MI_EINVAL;   // W, a, b: checked against their counts in prove_common; c == null: c = a o b
    const size_t wb = n_wires * 32, cb = n_constraints * 32;
    MI_TRY(mi_reserve(ctx, ctx->ws[16], wb + 3 * cb + 128));
    char *base = (char *)ctx->ws[16].p;
    const HostInputs host{W, a, b, c};
    return prove_common(ctx, pk, (mi_fr *)base, n_wires, (mi_fr *)(base + wb), (mi_fr *)(base + wb + cb), (mi_fr *)(base + wb + 2 * cb),
                        n_constraints, r, s, out, stats, &host);
}
}
