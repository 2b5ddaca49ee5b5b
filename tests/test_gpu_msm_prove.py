"""GPU parity (through the C-ABI): G1/G2 MSM and the full prove path against the oracle and the
committed golden fixtures, plus size-independent properties at BASELINE sizes."""
import json
import os
import numpy as np
import pytest
import pyref as P
import cref
from helpers import *
from gpu_common import load_binding

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
INF_G1 = None


@pytest.fixture(scope="module")
def ctx():
    B = load_binding()
    c = B.Context(0)
    yield c
    c.close()


def _jac_eq(got, want):
    """both normalised: equal limbs, or both infinity (Z == 0)"""
    k = got.shape[0] // 3
    if not want[2 * k:].any():
        return not got[2 * k:].any()
    return np.array_equal(got, want)


def test_msm_golden_json(ctx):
    with open(os.path.join(GOLD, "msm.json")) as f:
        g = json.load(f)
    for case in g["g1"]:
        pts = [None if p is None else (int(p[0], 16), int(p[1], 16)) for p in case["points"]]
        sc = [int(x, 16) for x in case["scalars"]]
        want = None if case["out"] is None else (int(case["out"][0], 16), int(case["out"][1], 16))
        assert g1_from_jac(ctx.msm_g1(g1_arr(pts), fr_arr(sc))) == want
    for case in g["g2"]:
        unflat = lambda p: None if p is None else ((int(p[0], 16), int(p[1], 16)), (int(p[2], 16), int(p[3], 16)))
        pts = [unflat(p) for p in case["points"]]
        sc = [int(x, 16) for x in case["scalars"]]
        assert g2_from_jac(ctx.msm_g2(g2_arr(pts), fr_arr(sc))) == unflat(case["out"])


def test_msm_golden_npz(ctx):
    for name in ("msm_g1_4096_uniform.npz", "msm_g1_4096_whir.npz"):
        z = np.load(os.path.join(GOLD, name))
        assert np.array_equal(ctx.msm_g1(z["points"], z["scalars"])[:8], z["out"])
    z = np.load(os.path.join(GOLD, "msm_g2_512_whir.npz"))
    assert np.array_equal(ctx.msm_g2(z["points"], z["scalars"])[:16], z["out"])


@pytest.mark.parametrize("n,dist", [(0, 0), (1, 0), (2, 1), (3, 0), (63, 1), (64, 0), (65, 1), (1000, 0), (4097, 1), (50000, 0), (200000, 1)])
def test_msm_g1_vs_oracle(ctx, n, dist):
    pts = cref.gen_g1(n, 77 + n); sc = cref.gen_scalars(n, 5 + n, dist)
    if n > 10:
        pts[3] = 0; sc[1] = fr_arr([P.R_MOD - 1])[0]; sc[2] = 0; pts[6] = pts[5]; sc[6] = sc[5]
        pts[8] = g1_arr([P.g1_neg(g1_pts(pts[7:8])[0])])[0]; sc[8] = sc[7]
    assert _jac_eq(ctx.msm_g1(pts, sc), cref.msm_g1(pts, sc))


@pytest.mark.parametrize("n,dist", [(1, 0), (17, 1), (300, 0), (5000, 1)])
def test_msm_g2_vs_oracle(ctx, n, dist):
    pts = cref.gen_g2(n, 177 + n); sc = cref.gen_scalars(n, 15 + n, dist)
    assert _jac_eq(ctx.msm_g2(pts, sc), cref.msm_g2(pts, sc))


def test_msm_edge_scalars(ctx):
    n = 200
    pts = cref.gen_g1(n, 1)
    zeros = np.zeros((n, 4), np.uint64)
    assert not ctx.msm_g1(pts, zeros)[8:].any()                       # all-zero scalars -> infinity
    ones = fr_arr([1] * n)
    assert g1_from_jac(ctx.msm_g1(pts, ones)) == P.ec_sum(P.F1, g1_pts(pts))   # plain point sum
    rm1 = fr_arr([P.R_MOD - 1] * n)
    assert g1_from_jac(ctx.msm_g1(pts, rm1)) == P.g1_neg(P.ec_sum(P.F1, g1_pts(pts)))
    allinf = np.zeros((n, 8), np.uint64)
    assert not ctx.msm_g1(allinf, ones)[8:].any()
    canon = cref.field_op(0, 5, cref.gen_scalars(n, 9, 1))
    assert np.array_equal(ctx.msm_g1(pts, canon, flags=1), cref.msm_g1(pts, canon, flags=1))


def test_msm_forced_small_windows_many_levels(ctx):
    """knobs turned down so that every level / window path runs at a size the oracle finishes fast"""
    lib = ctx.lib
    try:
        for (c, L1, L2, seg, G) in ((4, 3, 2, 2, 3), (7, 4, 3, 8, 5), (16, 8, 4, 64, 2), (11, 32, 16, 8, 1)):
            assert lib.mi_debug_set_msm_plan(ctx.h, c, L1, L2, seg, G) == 0
            for dist in (0, 1):
                pts = cref.gen_g1(3000, 31 + c); sc = cref.gen_scalars(3000, 41 + c, dist)
                assert _jac_eq(ctx.msm_g1(pts, sc), cref.msm_g1(pts, sc)), (c, dist)
            p2 = cref.gen_g2(300, 51 + c); s2 = cref.gen_scalars(300, 61 + c, 1)
            assert _jac_eq(ctx.msm_g2(p2, s2), cref.msm_g2(p2, s2)), c
    finally:
        assert lib.mi_debug_set_msm_plan(ctx.h, 0, 0, 0, 0, 0) == 0


@pytest.mark.parametrize("dist", [0, 1])
def test_msm_linearity_at_baseline_size(ctx, dist):
    """n = 2^23 pairs generated on the device: MSM(P, s) + MSM(P, t) == MSM(P, s + t), and
    MSM over the two halves adds up to the whole (the sharding identity config 5 relies on)."""
    B = load_binding()
    n = 1 << 23
    pts = ctx.gen_g1(n, 11); s = ctx.gen_scalars(n, 12, dist); t = ctx.gen_scalars(n, 13, 0)
    st = ctx.alloc(32 * n)
    ctx.field_op_dev(0, 0, st.ptr, s.ptr, t.ptr, n)
    ms, mt, mst = ctx.msm_g1_dev(pts.ptr, s.ptr, n), ctx.msm_g1_dev(pts.ptr, t.ptr, n), ctx.msm_g1_dev(pts.ptr, st.ptr, n)
    assert np.array_equal(B.g1_sum(np.stack([ms, mt])), mst)
    h = n // 2
    lo = ctx.msm_g1_dev(pts.ptr, s.ptr, h); hi = ctx.msm_g1_dev(pts.ptr + 64 * h, s.ptr + 32 * h, n - h)
    assert np.array_equal(B.g1_sum(np.stack([lo, hi])), ms)
    assert np.array_equal(cref.g1_sum(np.stack([lo, hi])), ms)   # the host combine agrees with the oracle's
    for d in (pts, s, t, st):
        d.free()


def _load_toy():
    z = np.load(os.path.join(GOLD, "prove_toy1000.npz"))
    pk = {k: z[k] for k in ("g1_a", "g1_b", "g1_k", "g1_z", "g2_b", "alpha1", "beta1", "delta1", "beta2", "delta2", "infinity_a", "infinity_b")}
    pk.update(log_n=int(z["log_n"]), nb_public=int(z["nb_public"]), nb_wires=int(z["nb_wires"]))
    return z, pk


def test_prove_golden_toy1000_bytes(ctx):
    """full proof of the 1000-constraint toy circuit: bit-exact proof bytes (gnark Proof.WriteTo layout)"""
    B = load_binding()
    z, pk = _load_toy()
    pkh = ctx.pk_load(pk)
    proof, stats = ctx.prove(pkh, z["W"], z["a"], z["b"], z["c"], z["r"], z["s"])
    assert np.array_equal(proof["ar"], z["ar"]) and np.array_equal(proof["bs"], z["bs"]) and np.array_equal(proof["krs"], z["krs"])
    assert B.proof_write(proof["raw"]) == bytes(z["proof_bytes"])
    with open(os.path.join(GOLD, "prove.json")) as f:
        assert B.proof_write(proof["raw"]).hex() == json.load(f)["proof_bytes"]
    assert stats["total_ms"] > 0
    # a second proof on the same resident key, different blinding: must match the oracle again
    r2, s2 = fr_arr([12345])[0], fr_arr([P.R_MOD - 7])[0]
    proof2, _ = ctx.prove(pkh, z["W"], z["a"], z["b"], z["c"], r2, s2)
    want2 = cref.prove(pk, z["W"], z["a"], z["b"], z["c"], r2, s2)
    assert B.proof_write(proof2["raw"]) == cref.proof_write(want2["raw"])
    ctx.pk_free(pkh)


@pytest.mark.parametrize("log_n,n_committed", [(10, 0), (13, 37), (16, 0)])
def test_prove_synthetic_vs_oracle(ctx, log_n, n_committed):
    """shape-faithful synthetic key + witness (infinity masks 10% / 50%, public wires, committed
    wires removed from K): GPU proof bytes == oracle proof bytes"""
    B = load_binding()
    n = 1 << log_n
    nb_wires, nb_public, n_constraints = n - 13, 41, n - 5
    pk = synthetic_pk(log_n, nb_wires, nb_public, 900 + log_n, n_committed=n_committed)
    W = cref.gen_scalars(nb_wires, 1, 1)
    a = cref.gen_scalars(n_constraints, 2, 1); b = cref.gen_scalars(n_constraints, 3, 0); c = cref.field_op(0, 2, a, b)
    r, s = cref.gen_scalars(2, 4, 0)
    pkh = ctx.pk_load(pk)
    got, _ = ctx.prove(pkh, W, a, b, c, r, s)
    want = cref.prove(pk, W, a, b, c, r, s)
    assert B.proof_write(got["raw"]) == cref.proof_write(want["raw"])
    ctx.pk_free(pkh)


@pytest.mark.parametrize("knob", [(17, 17, 17), (19, 18, 20), (1, 17, 1), (22, 1, 1), (1, 1, 1)])
def test_prove_with_fixed_base_tables_vs_oracle(ctx, knob):
    """mi_pk_load builds fixed-base window tables for the large MSM groups (A+K, B1+B2, Z); forced here at a size the
    oracle finishes in seconds, in every mix of table / generic groups: proof bytes == oracle proof bytes"""
    B = load_binding()
    log_n = 12
    n = 1 << log_n
    nb_wires, nb_public, n_constraints = n - 13, 41, n - 5
    pk = synthetic_pk(log_n, nb_wires, nb_public, 7700, n_committed=9)
    W = cref.gen_scalars(nb_wires, 1, 1)
    a = cref.gen_scalars(n_constraints, 2, 1); b = cref.gen_scalars(n_constraints, 3, 0); c = cref.field_op(0, 2, a, b)
    r, s = cref.gen_scalars(2, 4, 0)
    want = cref.proof_write(cref.prove(pk, W, a, b, c, r, s)["raw"])
    assert ctx.lib.mi_debug_set_prove_fixed_base(ctx.h, *knob) == 0
    try:
        pkh = ctx.pk_load(pk)
    finally:
        assert ctx.lib.mi_debug_set_prove_fixed_base(ctx.h, 0, 0, 0) == 0
    got, _ = ctx.prove(pkh, W, a, b, c, r, s)
    assert B.proof_write(got["raw"]) == want
    # device-resident key arrays owned by the caller (mi_pk_load_dev) stay untouched and usable next to the tables
    ctx.pk_free(pkh)
    assert ctx.lib.mi_debug_set_prove_fixed_base(ctx.h, 16, 0, 0) != 0 and ctx.lib.mi_debug_set_prove_fixed_base(ctx.h, 0, 23, 0) != 0


_CENSUS_CASE = {}


@pytest.mark.parametrize("knob", [(0, 0, 0), (19, 17, 20), (20, 18, 20)])
def test_prove_census_mix_at_2p20_vs_oracle(ctx, knob):
    """the witness mix tools/wire_census.py derives from the reference's circuit (mtUtilities.go:494-532 eq tables and
    matrix MLE all full-width; Merkle / STIR terms bytes and full-width; bits 1-3 %: MI_DIST_MIX, same generator on both sides) at
    N = 2^20, through the generic path and through fixed-base tables of the production widths: proof bytes == oracle proof bytes"""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import wire_census
    B = load_binding()
    pm = wire_census.census_mix_permille()
    assert sum(pm) < 500 and pm[2] == 0, "the census says: mostly full-width values, no 64-bit class"
    dist = B.dist_mix(*pm)
    log_n = 20
    n = 1 << log_n
    nb_wires, nb_public, n_constraints = n - 1000, 4097, n - 100
    if "case" not in _CENSUS_CASE:   # one key, one witness and one oracle proof for the three table plans
        pk = synthetic_pk(log_n, nb_wires, nb_public, 4400, n_committed=n >> 5)
        W = ctx.gen_scalars(nb_wires, 1, dist).download((nb_wires, 4)); a = ctx.gen_scalars(n_constraints, 2, dist).download((n_constraints, 4))
        assert np.array_equal(W[:4000], cref.gen_scalars(4000, 1, dist))
        b = cref.gen_scalars(n_constraints, 3, 0); c = cref.field_op(0, 2, a, b)
        r, s = cref.gen_scalars(2, 4, 0)
        _CENSUS_CASE["case"] = (pk, W, a, b, r, s, cref.proof_write(cref.prove(pk, W, a, b, c, r, s)["raw"]))
    pk, W, a, b, r, s, want = _CENSUS_CASE["case"]
    assert ctx.lib.mi_debug_set_prove_fixed_base(ctx.h, *knob) == 0
    try:
        pkh = ctx.pk_load(pk)
    finally:
        assert ctx.lib.mi_debug_set_prove_fixed_base(ctx.h, 0, 0, 0) == 0
    got, st = ctx.prove(pkh, W, a, b, None, r, s)      # c formed on the device, as on the benchmarked path
    assert B.proof_write(got["raw"]) == want
    if knob[0]:   # ~3/4 of the wire scalars are full-width: about 0.76 x 14 digits each against the WHIR mix's ~4
        assert st["g1_level1_additions"] > 8 * nb_wires
    ctx.pk_free(pkh)


def test_prove_rejects_mismatched_inputs(ctx):
    z, pk = _load_toy()
    bad = dict(pk); bad["g1_a"] = pk["g1_a"][:-1]
    B = load_binding()
    with pytest.raises(B.MiError):
        ctx.pk_load(bad)
    pkh = ctx.pk_load(pk)
    with pytest.raises(B.MiError):
        ctx.prove(pkh, z["W"][:-1], z["a"], z["b"], z["c"], z["r"], z["s"])
    ctx.pk_free(pkh)


def test_msm_sharding_identity_at_2p26(ctx):
    """maximum single-GPU size of BASELINE configs[2]/[4] (2^26 pairs): the two halves add up to the whole"""
    B = load_binding()
    n = 1 << 26
    pts = ctx.gen_g1(n, 21); s = ctx.gen_scalars(n, 22, 1)
    whole = ctx.msm_g1_dev(pts.ptr, s.ptr, n)
    h = n // 2 + 12345
    lo = ctx.msm_g1_dev(pts.ptr, s.ptr, h); hi = ctx.msm_g1_dev(pts.ptr + 64 * h, s.ptr + 32 * h, n - h)
    assert np.array_equal(B.g1_sum(np.stack([lo, hi])), whole)
    assert whole[8:].any()
    for d in (pts, s):
        d.free()


def test_msm_g2_repeated_and_opposite_points(ctx):
    """buckets that see P + P (doubling) and P + (-P) (cancellation) on the G2 path"""
    n = 400
    pts = cref.gen_g2(n, 321); sc = cref.gen_scalars(n, 322, 1)
    for i in range(0, 60, 3):
        pts[i + 1] = pts[i]; sc[i + 1] = sc[i]                                   # same point, same scalar: doubling in a bucket
        pts[i + 2] = g2_arr([P.g2_neg(g2_pts(pts[i:i + 1])[0])])[0]; sc[i + 2] = sc[i]   # opposite point: cancellation
    pts[77] = 0
    assert _jac_eq(ctx.msm_g2(pts, sc), cref.msm_g2(pts, sc))
    allsame = np.repeat(pts[:1], 64, axis=0); ones = fr_arr([1] * 64)
    assert g2_from_jac(ctx.msm_g2(allsame, ones)) == P.g2_mul(g2_pts(pts[:1])[0], 64)


@pytest.mark.parametrize("n", [1, 300, 5000, 70000])
def test_pedersen_commit_and_pok_vs_oracle(ctx, n):
    """BSB22 Pedersen Commit / ProveKnowledge through the device-resident key"""
    B = load_binding()
    basis = cref.gen_g1(n + 5, 500 + n); bes = cref.gen_g1(n + 5, 600 + n); vals = cref.gen_scalars(n, 700 + n, 1)
    pk = ctx.pedersen_pk_load(basis, bes)
    assert np.array_equal(ctx.pedersen_commit(pk, vals), cref.pedersen_msm(basis, vals))
    assert np.array_equal(ctx.pedersen_commit(pk, vals, knowledge=True), cref.pedersen_msm(bes, vals))
    with pytest.raises(B.MiError):
        ctx.pedersen_commit(pk, cref.gen_scalars(n + 6, 1, 0))      # more values than basis points
    ctx.pedersen_pk_free(pk)


def test_proof_with_commitment_is_196_bytes_and_matches_oracle(ctx):
    """the WHIR circuit's proof carries one BSB22 commitment + its PoK: Ar | Bs | Krs | 1 | commitment | pok = 196 bytes"""
    B = load_binding()
    z, pk = _load_toy()
    pkh = ctx.pk_load(pk)
    proof, _ = ctx.prove(pkh, z["W"], z["a"], z["b"], z["c"], z["r"], z["s"])
    basis = cref.gen_g1(40, 1); bes = cref.gen_g1(40, 2); vals = cref.gen_scalars(40, 3, 1)
    ppk = ctx.pedersen_pk_load(basis, bes)
    com = ctx.pedersen_commit(ppk, vals); pok = B.pedersen_fold(ctx.pedersen_commit(ppk, vals, knowledge=True).reshape(1, 8), cref.gen_scalars(1, 4, 0)[0])
    got = B.proof_write(proof["raw"], commitments=com.reshape(1, 8).copy(), pok=pok.copy())
    want_pok = cref.pedersen_fold(cref.pedersen_msm(bes, vals).reshape(1, 8), cref.gen_scalars(1, 4, 0)[0])
    assert len(got) == 196 and got == cref.proof_write(proof["raw"], commitments=cref.pedersen_msm(basis, vals).reshape(1, 8).copy(), pok=want_pok.copy())
    ctx.pedersen_pk_free(ppk); ctx.pk_free(pkh)


def test_batch_scalar_mul_vs_oracle(ctx):
    """fixed-base batch scalar multiplication (groth16.Setup's BatchScalarMultiplicationG1/G2)"""
    n = 20000
    sc = cref.gen_scalars(n, 95, 0); sc[0] = 0; sc[1] = fr_arr([1])[0]; sc[2] = fr_arr([P.R_MOD - 1])[0]; sc[3:40] = cref.gen_scalars(37, 96, 1)
    base = cref.gen_g1(1, 97)[0]
    assert np.array_equal(ctx.batch_scalar_mul(base, sc), cref.batch_scalar_mul(base, sc))
    gen = g1_arr([P.G1_GEN])[0]
    assert np.array_equal(ctx.batch_scalar_mul(gen, sc[:500]), cref.batch_scalar_mul(gen, sc[:500]))
    b2 = cref.gen_g2(1, 98)[0]
    assert np.array_equal(ctx.batch_scalar_mul(b2, sc[:3000], g2=True), cref.batch_scalar_mul(b2, sc[:3000], g2=True))
    # a toy Setup row: the points of pk.G1.Z are zdt * tau^i * G1 -> consecutive quotients are tau
    assert ctx.batch_scalar_mul(gen, sc[:0]).shape == (0, 8)


@pytest.mark.parametrize("log_n", [23, 25, 26])
def test_full_size_prove_equals_composition_of_primitives(ctx, log_n):
    """BASELINE configs[1] size (N = 2^23, WHIR scalar mix), N = 2^25 and configs[2] size (2^26): the fused multi-stream
    prove -- fixed-base tables for all three MSM groups at 2^23, for Z and B at 2^25, for Z alone at 2^26 (the others do not
    fit the budget and stay generic) -- must equal
    the proof assembled from the separately tested primitives (computeH, four generic G1 MSMs, one generic G2 MSM run one
    by one on the same device arrays) plus O(1) point operations done by the oracle."""
    B = load_binding()
    N = 1 << log_n
    # the module's context has run every earlier test: its grow-only workspaces go back first (mi_ctx_trim), so that the sizes that
    # want most of the GPU do not inherit them (round 3: an out-of-memory at 2^26 after 70 tests' worth of workspaces)
    ctx.trim()
    led = ctx.mem_ledger()
    assert sum(v for k, v in led.items() if k.startswith("ctx_")) == 0, led
    nb_wires, nb_public, n_constraints = N - 1000, 4097, N - 100
    rng = np.random.default_rng(7)
    inf_a = (rng.integers(0, 100, nb_wires) < 10).astype(np.uint8); inf_b = (rng.integers(0, 100, nb_wires) < 50).astype(np.uint8)
    na, nb, nk = int((inf_a == 0).sum()), int((inf_b == 0).sum()), nb_wires - nb_public
    g1a, g1b, g1k, g1z, g2b = ctx.gen_g1(na, 1), ctx.gen_g1(nb, 2), ctx.gen_g1(nk, 3), ctx.gen_g1(N, 4), ctx.gen_g2(nb, 5)
    small = cref.gen_g1(3, 6); small2 = cref.gen_g2(2, 7)
    pk = {"log_n": log_n, "nb_public": nb_public, "nb_wires": nb_wires, "g1_a": (g1a.ptr, na), "g1_b": (g1b.ptr, nb), "g1_k": (g1k.ptr, nk),
          "g1_z": (g1z.ptr, N), "g2_b": (g2b.ptr, nb), "alpha1": small[0], "beta1": small[1], "delta1": small[2], "beta2": small2[0],
          "delta2": small2[1], "infinity_a": inf_a, "infinity_b": inf_b}
    pkh = ctx.pk_load(pk, device_points=True)
    W = ctx.gen_scalars(nb_wires, 8, 1); a = ctx.gen_scalars(n_constraints, 9, 1); b = ctx.gen_scalars(n_constraints, 10, 0)
    c = ctx.alloc(32 * n_constraints); ctx.field_op_dev(0, 2, c.ptr, a.ptr, b.ptr, n_constraints)
    r, s = cref.gen_scalars(2, 11, 0)
    proof, _ = ctx.prove(pkh, W.ptr, a.ptr, b.ptr, c.ptr, r, s, device=True, n_wires=nb_wires, n_constraints=n_constraints)
    # the same proof from the primitives, one at a time
    h = ctx.alloc(32 * N); ctx.compute_h_dev(log_n, a.ptr, b.ptr, c.ptr, n_constraints, h.ptr)
    Wh = W.download((nb_wires, 4))
    wa, wb, wk = ctx.to_dev(Wh[inf_a == 0]), ctx.to_dev(Wh[inf_b == 0]), ctx.to_dev(Wh[nb_public:])
    m_a, m_b1, m_k = ctx.msm_g1_dev(g1a.ptr, wa.ptr, na), ctx.msm_g1_dev(g1b.ptr, wb.ptr, nb), ctx.msm_g1_dev(g1k.ptr, wk.ptr, nk)
    m_z, m_b2 = ctx.msm_g1_dev(g1z.ptr, h.ptr, N - 1), ctx.msm_g2_dev(g2b.ptr, wb.ptr, nb)
    rc, sc = fr_vals(r)[0], fr_vals(s)[0]
    aff = lambda j: j[:8]
    add = lambda x, y: cref.g1_add(x.reshape(1, 8), y.reshape(1, 8))[0]
    ar = add(add(aff(m_a), small[0]), cref.g1_scalar_mul(small[2], rc))
    bs1 = add(add(aff(m_b1), small[1]), cref.g1_scalar_mul(small[2], sc))
    krs = add(add(aff(m_k), aff(m_z)), cref.g1_scalar_mul(small[2], (-rc * sc) % P.R_MOD))
    krs = add(add(krs, cref.g1_scalar_mul(ar, sc)), cref.g1_scalar_mul(bs1, rc))
    bs = cref.g2_add(cref.g2_add(m_b2[:16].reshape(1, 16), small2[0].reshape(1, 16)), cref.g2_scalar_mul(small2[1], sc).reshape(1, 16))[0]
    assert np.array_equal(proof["ar"], ar) and np.array_equal(proof["krs"], krs) and np.array_equal(proof["bs"], bs)
    ctx.pk_free(pkh)
    for d in (g1a, g1b, g1k, g1z, g2b, W, a, b, c, h, wa, wb, wk):
        d.free()


@pytest.mark.parametrize("log_n,nb_wires,nb_public,n_constraints,ia,ib", [
    (0, 1, 1, 1, 0, 0),          # a single constraint, only the ONE wire: every MSM but Z... is trivial, Z has 0 points
    (1, 3, 3, 2, 100, 100),      # no private wires, every A and B point at infinity
    (3, 9, 2, 5, 0, 100),        # B empty: Bs = beta + s*delta
    (5, 40, 7, 32, 100, 0),      # A empty, full domain of constraints
    (9, 300, 300, 512, 30, 30),  # K empty (all wires public)
])
def test_prove_degenerate_shapes_vs_oracle(ctx, log_n, nb_wires, nb_public, n_constraints, ia, ib):
    """empty / all-infinity / public-only shapes: the fused path must still equal the oracle byte for byte"""
    B = load_binding()
    pk = synthetic_pk(log_n, nb_wires, nb_public, 4000 + log_n, inf_a_pct=ia, inf_b_pct=ib)
    W = cref.gen_scalars(nb_wires, 1, 1)
    a = cref.gen_scalars(n_constraints, 2, 1); b = cref.gen_scalars(n_constraints, 3, 0); c = cref.field_op(0, 2, a, b)
    r, s = cref.gen_scalars(2, 4, 0)
    pkh = ctx.pk_load(pk)
    got, _ = ctx.prove(pkh, W, a, b, c, r, s)
    want = cref.prove(pk, W, a, b, c, r, s)
    assert B.proof_write(got["raw"]) == cref.proof_write(want["raw"])
    ctx.pk_free(pkh)


@pytest.mark.parametrize("n,dist,c,chunk,gbits", [(1, 0, 17, 0, 0), (300, 1, 17, 64, 6), (5000, 0, 18, 1000, 15), (70000, 1, 20, 0, 0),
                                                  (200000, 0, 22, 0, 9), (513, 1, 19, 100, 12), (66000, 0, 21, 4096, 10)])
def test_fixed_base_msm_g1_vs_oracle(ctx, n, dist, c, chunk, gbits):
    """fixed-base path: window copies 2^(c*w)*P built on the device, one bucket set, two-pass sort (pass 1 staged through
    LDS per slice of 512 scalars, group width and chunk size forced through the knobs)"""
    assert ctx.lib.mi_debug_set_msm_chunk(ctx.h, chunk) == 0 and ctx.lib.mi_debug_set_msm_group_bits(ctx.h, gbits) == 0
    try:
        pts = cref.gen_g1(n, 1300 + n); sc = cref.gen_scalars(n, 1400 + n, dist)
        if n > 10:
            pts[3] = 0; sc[1] = fr_arr([P.R_MOD - 1])[0]; sc[2] = 0; pts[6] = pts[5]; sc[6] = sc[5]
        dp, ds = ctx.to_dev(pts), ctx.to_dev(sc)
        pre = ctx.msm_precompute(dp.ptr, n, c)
        nwin = (256 + c - 1) // c
        # the copies themselves: pre[w][i] == 2^(c*w) * P_i
        got = pre.download((nwin * n, 8))
        i = 0 if n < 10 else 9
        assert np.array_equal(got[i], pts[i]) and np.array_equal(got[(nwin - 1) * n + i], cref.g1_scalar_mul(pts[i], 1 << (c * (nwin - 1))))
        assert _jac_eq(ctx.msm_fixed_dev(pre.ptr, ds.ptr, n, c), cref.msm_g1(pts, sc))
        for d in (dp, ds, pre):
            d.free()
    finally:
        assert ctx.lib.mi_debug_set_msm_chunk(ctx.h, 0) == 0 and ctx.lib.mi_debug_set_msm_group_bits(ctx.h, 0) == 0
    assert ctx.lib.mi_debug_set_msm_group_bits(ctx.h, 5) != 0 and ctx.lib.mi_debug_set_msm_group_bits(ctx.h, 16) != 0


def test_fixed_base_msm_g2_vs_oracle(ctx):
    n, c = 3000, 19
    pts = cref.gen_g2(n, 1500); sc = cref.gen_scalars(n, 1501, 1)
    dp, ds = ctx.to_dev(pts), ctx.to_dev(sc)
    pre = ctx.msm_precompute(dp.ptr, n, c, g2=True)
    assert _jac_eq(ctx.msm_fixed_dev(pre.ptr, ds.ptr, n, c, g2=True), cref.msm_g2(pts, sc))
    for d in (dp, ds, pre):
        d.free()


def test_fixed_base_equals_generic_at_baseline_size(ctx):
    """2^23 uniform pairs: the fixed-base path (c = 22, 12 digits) and the generic path (c = 16) agree"""
    n, c = 1 << 23, 22
    pts = ctx.gen_g1(n, 31); sc = ctx.gen_scalars(n, 32, 0)
    want = ctx.msm_g1_dev(pts.ptr, sc.ptr, n); t_gen = ctx.stats()["total_ms"]
    pre = ctx.msm_precompute(pts.ptr, n, c)
    got = ctx.msm_fixed_dev(pre.ptr, sc.ptr, n, c); got = ctx.msm_fixed_dev(pre.ptr, sc.ptr, n, c); t_fix = ctx.stats()["total_ms"]
    print(f"generic {t_gen:.2f} ms, fixed-base {t_fix:.2f} ms")
    assert np.array_equal(got, want)
    for d in (pts, sc, pre):
        d.free()


@pytest.mark.parametrize("n,dist", [(1 << 20, 1), ((1 << 21) + 12345, 0)])
def test_generic_msm_two_pass_and_one_pass_sorts_agree_with_oracle(ctx, n, dist):
    """generic MSMs with c = 16 (n >= 2^20) sort through the LDS-staged two-pass sort with the window folded into the key;
    the one-pass counting sort stays behind a knob: both against the oracle, G1 and G2"""
    pts = cref.gen_g1(n, 2100 + dist); sc = cref.gen_scalars(n, 2200 + dist, dist)
    pts[3] = 0; sc[1] = fr_arr([P.R_MOD - 1])[0]; sc[2] = 0; pts[6] = pts[5]; sc[6] = sc[5]
    want = cref.msm_g1(pts, sc)
    dp, ds = ctx.to_dev(pts), ctx.to_dev(sc)
    n2 = 1 << 20
    p2 = ctx.gen_g2(n2, 77); want2 = cref.msm_g2(p2.download((n2, 16)), sc[:n2])
    try:
        for one_pass in (0, 1):
            assert ctx.lib.mi_debug_set_msm_one_pass_sort(ctx.h, one_pass) == 0
            assert np.array_equal(ctx.msm_g1_dev(dp.ptr, ds.ptr, n), want), one_pass
            assert np.array_equal(ctx.msm_g2_dev(p2.ptr, ds.ptr, n2), want2), one_pass
    finally:
        assert ctx.lib.mi_debug_set_msm_one_pass_sort(ctx.h, 0) == 0
    for d in (dp, ds, p2):
        d.free()


def test_limb29_level1_kernel_on_and_off_agree_with_oracle(ctx):
    """the G1 level-1 accumulation in nine 29-bit limbs with partial sums in the R' form (default, 1), with standard-form partial
    sums (2), and in 8 x 32-bit limbs (0), generic and fixed-base keys: proof bytes and a generic MSM with repeated / opposite /
    infinity points (100 copies of one point: equal partial sums, the doubling path of the R'-form addition) against the oracle"""
    B = load_binding()
    log_n = 15
    N = 1 << log_n
    pk = synthetic_pk(log_n, N - 50, 300, 5151, n_committed=9)
    W = cref.gen_scalars(N - 50, 1, 1); a = cref.gen_scalars(N - 10, 2, 1); b = cref.gen_scalars(N - 10, 3, 0); c = cref.field_op(0, 2, a, b)
    r, s = cref.gen_scalars(2, 4, 0)
    want = cref.proof_write(cref.prove(pk, W, a, b, c, r, s)["raw"])
    n = 1 << 17
    pts = cref.gen_g1(n, 88); sc = cref.gen_scalars(n, 89, 1)
    pts[3] = 0; pts[6] = pts[5]; sc[6] = sc[5]; pts[8] = g1_arr([P.g1_neg(g1_pts(pts[7:8])[0])])[0]; sc[8] = sc[7]; pts[100:200] = pts[99]
    want_msm = cref.msm_g1(pts, sc)
    n2 = 1 << 15
    p2 = cref.gen_g2(n2, 90); s2 = cref.gen_scalars(n2, 91, 1)
    p2[3] = 0; p2[6] = p2[5]; s2[6] = s2[5]; p2[8] = g2_arr([P.g2_neg(g2_pts(p2[7:8])[0])])[0]; s2[8] = s2[7]; p2[40:60] = p2[39]
    want_msm2 = cref.msm_g2(p2, s2)
    try:
        for on in (1, 2, 0, 3):   # 3: the default arithmetic with the level-1 kernel's two-waves-per-SIMD build
            assert ctx.lib.mi_debug_set_msm_limb29(ctx.h, 1 if on == 3 else on) == 0
            assert ctx.lib.mi_debug_set_msm_l1_waves(ctx.h, 2 if on == 3 else 3) == 0
            for knob in ((0, 0, 0), (17, 18, 17)):
                assert ctx.lib.mi_debug_set_prove_fixed_base(ctx.h, *knob) == 0
                pkh = ctx.pk_load(pk)
                got, _ = ctx.prove(pkh, W, a, b, c, r, s)
                ctx.pk_free(pkh)
                assert B.proof_write(got["raw"]) == want, (on, knob)
            assert np.array_equal(ctx.msm_g1(pts, sc), want_msm), on
            assert np.array_equal(ctx.msm_g2(p2, s2), want_msm2), on
    finally:
        assert ctx.lib.mi_debug_set_msm_limb29(ctx.h, 1) == 0 and ctx.lib.mi_debug_set_prove_fixed_base(ctx.h, 0, 0, 0) == 0
        assert ctx.lib.mi_debug_set_msm_l1_waves(ctx.h, 3) == 0


@pytest.mark.parametrize("g2,n", [(False, 65535), (False, 65536), (False, 65537), (False, 100003), (True, 16383), (True, 16384), (True, 16385), (True, 20011)])
def test_msm_around_the_limb29_thresholds(ctx, g2, n):
    """the public MSM entry points switch to the 29-bit level-1 kernels (bases converted into scratch) at 2^16 pairs (G1) and
    2^14 (G2): sizes on both sides of the switch, with infinity / repeated / opposite points, against the oracle"""
    sc = cref.gen_scalars(n, 3300 + n, 1)
    if g2:
        pts = cref.gen_g2(n, 3400 + n); pts[3] = 0; pts[6] = pts[5]; sc[6] = sc[5]; pts[8] = g2_arr([P.g2_neg(g2_pts(pts[7:8])[0])])[0]; sc[8] = sc[7]
        assert np.array_equal(ctx.msm_g2(pts, sc), cref.msm_g2(pts, sc))
    else:
        pts = cref.gen_g1(n, 3400 + n); pts[3] = 0; pts[6] = pts[5]; sc[6] = sc[5]; pts[8] = g1_arr([P.g1_neg(g1_pts(pts[7:8])[0])])[0]; sc[8] = sc[7]
        assert np.array_equal(ctx.msm_g1(pts, sc), cref.msm_g1(pts, sc))


@pytest.mark.parametrize("g2,n,c", [(False, 1, 17), (False, 1000, 20), (False, 4099, 19), (True, 777, 17)])
def test_window_tables_batched_and_per_point_conversions_are_identical(ctx, g2, n, c):
    """mi_msm_precompute: the tables built with one inversion per 16 points per window (default) and with one per point have the same
    bytes -- infinity bases, a ragged tail (n not a multiple of 16), both curves; a row of the table against the oracle's scalar
    multiplication"""
    pts = cref.gen_g2(n, 2100 + n) if g2 else cref.gen_g1(n, 2000 + n)
    if n > 20:
        pts[0] = 0; pts[17] = 0; pts[n - 1] = 0; pts[5] = pts[4]
    dp = ctx.to_dev(pts)
    nwin = (256 + c - 1) // c
    tabs = []
    try:
        for on in (1, 0):
            assert ctx.lib.mi_debug_set_msm_precompute_batched(ctx.h, on) == 0
            pre = ctx.msm_precompute(dp.ptr, n, c, g2=g2)
            tabs.append(pre.download((nwin * n, 16 if g2 else 8)))
            pre.free()
    finally:
        assert ctx.lib.mi_debug_set_msm_precompute_batched(ctx.h, 1) == 0
    assert np.array_equal(tabs[0], tabs[1])
    assert np.array_equal(tabs[0][:n], pts)
    if not g2:
        i = min(3, n - 1)
        assert np.array_equal(tabs[0][(nwin - 1) * n + i], cref.g1_scalar_mul(pts[i], 1 << (c * (nwin - 1))))
    dp.free()


@pytest.mark.parametrize("n,dist", [(60000, 1), (300000, 0), (1 << 20, 1)])
def test_msm_exact_and_worst_case_level_counts_agree_with_oracle(ctx, n, dist):
    """the item levels of an MSM are counted from the fullest bucket of its own sort (one 4-byte read-back, default) or from the worst
    case (every entry in one bucket): same sums, G1 and G2, small (one-pass sort) and large (two-pass sort) MSMs, skewed and uniform
    scalars -- with a giant bucket (a third of the scalars equal 1) in the skewed ones"""
    pts = cref.gen_g1(n, 7100 + n); sc = cref.gen_scalars(n, 7101 + n, dist)
    if dist:
        sc[::3] = fr_arr([1])[0]
    n2 = min(n, 40000)
    p2 = cref.gen_g2(n2, 7102); s2 = sc[:n2]
    want1, want2 = cref.msm_g1(pts, sc), cref.msm_g2(p2, s2)
    try:
        for on in (0, 1, 0):
            assert ctx.lib.mi_debug_set_msm_bound_levels(ctx.h, on) == 0
            assert np.array_equal(ctx.msm_g1(pts, sc), want1), on
            assert np.array_equal(ctx.msm_g2(p2, s2), want2), on
    finally:
        assert ctx.lib.mi_debug_set_msm_bound_levels(ctx.h, 0) == 0


@pytest.mark.parametrize("rounds", [1, 2, 3, 4])
def test_batch_affine_level1_rounds_agree_with_oracle(ctx, rounds):
    """the experimental batch-affine level 1 (mi_debug_set_msm_batch_affine, csrc/msm_ba_g1.cuh; off by default): buckets of many
    entries (small windows) so that it engages, with repeated points (doublings: the slots that leave the batch), opposite pairs
    (cancellation inside an item), points at infinity, ragged items; and every pair an equal-point addition (one point, one scalar)"""
    lib = ctx.lib
    n = 1 << 16
    pts = cref.gen_g1(n, 188); sc = cref.gen_scalars(n, 189, 1)
    pts[3] = 0; pts[6] = pts[5]; sc[6] = sc[5]; pts[8] = g1_arr([P.g1_neg(g1_pts(pts[7:8])[0])])[0]; sc[8] = sc[7]; pts[100:300] = pts[99]
    pts[1000:1100] = 0
    want = cref.msm_g1(pts, sc)
    same = cref.gen_g1(n, 190); same[:] = same[0]
    ssc = cref.gen_scalars(n, 191, 0); ssc[:] = ssc[0]
    want_same = cref.msm_g1(same, ssc)
    try:
        assert lib.mi_debug_set_msm_batch_affine(ctx.h, rounds) == 0
        for c in (8, 9):
            assert lib.mi_debug_set_msm_plan(ctx.h, c, 0, 0, 0, 0) == 0
            assert _jac_eq(ctx.msm_g1(pts, sc), want), (rounds, c)
        assert lib.mi_debug_set_msm_plan(ctx.h, 5, 0, 0, 0, 0) == 0
        assert _jac_eq(ctx.msm_g1(same, ssc), want_same), rounds
    finally:
        assert lib.mi_debug_set_msm_batch_affine(ctx.h, 0) == 0 and lib.mi_debug_set_msm_plan(ctx.h, 0, 0, 0, 0, 0) == 0


def test_published_alt_bn128_vectors_through_the_hip_path(ctx):
    """EIP-196 ecAdd / ecMul and EIP-197's G2 generator (the vectors tests/test_oracle.py pins both oracles with) through the device
    code: point addition kernels, the MSMs (level-1 mixed additions, doubling fallback, bucket reduce), the batch scalar multiplication"""
    from test_oracle import EIP196_ADD, EIP196_MUL, EIP196_DOUBLE_G, EIP196_TRIPLE_G, EIP197_G2
    B = load_binding()
    a, b, c = EIP196_ADD
    assert g1_pts(ctx.ec_add(g1_arr([a, P.G1_GEN, P.G1_GEN]), g1_arr([b, P.G1_GEN, EIP196_DOUBLE_G]))) == [c, EIP196_DOUBLE_G, EIP196_TRIPLE_G]
    assert g1_from_jac(ctx.msm_g1(g1_arr([a, b]), fr_arr([1, 1]))) == c
    pt, k, want = EIP196_MUL
    assert g1_from_jac(ctx.msm_g1(g1_arr([pt]), fr_arr([k]))) == want
    assert g1_pts(ctx.batch_scalar_mul(g1_arr([pt])[0], fr_arr([k, 1])))[0] == want
    assert g1_pts(ctx.batch_scalar_mul(g1_arr([P.G1_GEN])[0], fr_arr([2, 3, P.R_MOD - 1]))) == [EIP196_DOUBLE_G, EIP196_TRIPLE_G, P.g1_neg(P.G1_GEN)]
    # r G = infinity through an MSM that is all one point: (r - 1) G + G, on both curves
    assert g1_from_jac(ctx.msm_g1(g1_arr([P.G1_GEN, P.G1_GEN]), fr_arr([P.R_MOD - 1, 1]))) is None
    assert g2_from_jac(ctx.msm_g2(g2_arr([EIP197_G2, EIP197_G2]), fr_arr([P.R_MOD - 1, 1]))) is None
    assert g2_pts(ctx.batch_scalar_mul(g2_arr([EIP197_G2])[0], fr_arr([P.R_MOD - 1]), g2=True)) == [P.g2_neg(EIP197_G2)]


def test_trim_gives_the_workspaces_back_and_the_next_proof_is_the_same(ctx):
    """mi_ctx_trim / mi_prover_trim: an idle context (pool) frees its grow-only workspaces; proving goes on with the same bytes"""
    B = load_binding()
    log_n = 14
    N = 1 << log_n
    pk = synthetic_pk(log_n, N - 50, 33, 6100, n_committed=9)
    W = cref.gen_scalars(N - 50, 1, 1); a = cref.gen_scalars(N - 10, 2, 1); b = cref.gen_scalars(N - 10, 3, 0); c = cref.field_op(0, 2, a, b)
    r, s = cref.gen_scalars(2, 4, 0)
    want = cref.proof_write(cref.prove(pk, W, a, b, c, r, s)["raw"])
    pkh = ctx.pk_load(pk)
    assert B.proof_write(ctx.prove(pkh, W, a, b, c, r, s)[0]["raw"]) == want
    before = ctx.mem_ledger(pkh)
    assert before["ctx_msm"] > 0 and before["ctx_ntt_vectors"] > 0
    ctx.trim()
    after = ctx.mem_ledger(pkh)
    assert all(after[k] == 0 for k in after if k.startswith("ctx_")), after
    assert after["key_bases"] == before["key_bases"] and after["key_tables"] == before["key_tables"]   # keys are not workspaces
    assert B.proof_write(ctx.prove(pkh, W, a, b, c, r, s)[0]["raw"]) == want
    assert np.array_equal(ctx.ntt(a[:1 << 12], 12, 3), cref.ntt(a[:1 << 12], 12, 3))
    pool = B.Prover(0, 2)
    t = pool.submit(pkh, W, a, b, None, r, s)
    with pytest.raises(B.MiError, match="idle"):
        pool.trim()   # (a job is queued or running; if it has finished already the trim simply succeeds: then raise by hand)
        raise B.MiError("idle: the job had finished")
    assert B.proof_write(pool.wait(t)[0]["raw"]) == want
    pool.trim()
    assert all(v == 0 for k, v in pool.ctx(0).mem_ledger().items() if k.startswith("ctx_"))
    assert B.proof_write(pool.wait(pool.submit(pkh, W, a, b, c, r, s))[0]["raw"]) == want
    pool.close()
    ctx.pk_free(pkh)


@pytest.mark.parametrize("n,dist", [(1 << 15, 1), (70001, 0), (200000, 1)])
def test_pedersen_full_length_commitments_use_the_window_tables_and_match_oracle(ctx, n, dist):
    """a Pedersen key of >= 2^15 points carries fixed-base window tables of Basis and BasisExpSigma; the full-length call (gnark's: as
    many values as basis elements) goes through them, a shorter one through the plain bases: both equal the oracle"""
    basis = cref.gen_g1(n, 510 + n); bes = cref.gen_g1(n, 610 + n); vals = cref.gen_scalars(n, 710 + n, dist)
    vals[0] = 0; vals[1] = fr_arr([1])[0]; vals[2] = fr_arr([P.R_MOD - 1])[0]; basis[5] = 0; bes[7] = bes[6]; vals[7] = vals[6]
    pk = ctx.pedersen_pk_load(basis, bes)
    assert np.array_equal(ctx.pedersen_commit(pk, vals), cref.pedersen_msm(basis, vals))
    assert np.array_equal(ctx.pedersen_commit(pk, vals, knowledge=True), cref.pedersen_msm(bes, vals))
    assert np.array_equal(ctx.pedersen_commit(pk, vals[:n - 3]), cref.pedersen_msm(basis[:n - 3], vals[:n - 3]))
    ctx.pedersen_pk_free(pk)
