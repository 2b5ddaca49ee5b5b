"""GPU parity for the round-5 code paths behind mi_debug_set_knob: the level-1 kernels as 1 / 2 / 4-wave workgroups (G1 both builds, G2),
and the finisher (one launch that ends the item levels) on / off and with small thresholds, in every partial-sum representation --
all against the oracle, with the inputs that make the levels interesting (a giant bucket, repeated and opposite points, infinities)."""
import numpy as np
import pytest
import pyref as P
import cref
from helpers import *
from gpu_common import load_binding

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    B = load_binding()
    c = B.Context(0)
    yield c
    c.close()


def _reset(ctx):
    for k, v in (("l1_wg", 4), ("g2_wg", 1), ("l1_waves", 3), ("z_waves", 0), ("finisher", 1), ("finisher_max", 0), ("plain_scatter", 0), ("count_per", 0),
                 ("g1_grid_per_cu", 0), ("g2_grid_per_cu", 0), ("finisher_min_level", 2), ("z_count_fused", 1), ("flat_item_l1", 0), ("dense_item_l1", 0)):
        ctx.set_knob(k, v)
    assert ctx.lib.mi_debug_set_msm_limb29(ctx.h, 1) == 0 and ctx.lib.mi_debug_set_msm_plan(ctx.h, 0, 0, 0, 0, 0) == 0


def _skewed(n, seed, g2=False):
    """WHIR-mix scalars with a giant bucket (a third equal 1), repeated points, an opposite pair, infinities"""
    pts = (cref.gen_g2 if g2 else cref.gen_g1)(n, seed); sc = cref.gen_scalars(n, seed + 1, 1)
    sc[::3] = fr_arr([1])[0]
    pts[3] = 0; pts[6] = pts[5]; sc[6] = sc[5]
    neg = g2_arr([P.g2_neg(g2_pts(pts[7:8])[0])])[0] if g2 else g1_arr([P.g1_neg(g1_pts(pts[7:8])[0])])[0]
    pts[8] = neg; sc[8] = sc[7]
    pts[100:200] = pts[99]
    return pts, sc


def test_unknown_knob_is_rejected(ctx):
    B = load_binding()
    with pytest.raises(B.MiError):
        ctx.set_knob("no_such_knob", 1)
    with pytest.raises(B.MiError):
        ctx.set_knob("l1_wg", 3)


@pytest.mark.parametrize("wg", [1, 2, 4])
def test_level1_workgroup_sizes_agree_with_oracle(ctx, wg):
    n = (1 << 17) + 77
    pts, sc = _skewed(n, 9100)
    n2 = (1 << 15) + 13
    p2, s2 = _skewed(n2, 9200, g2=True)
    want1, want2 = cref.msm_g1(pts, sc), cref.msm_g2(p2, s2)
    try:
        for waves in (3, 2):
            ctx.set_knob("l1_wg", wg); ctx.set_knob("g2_wg", wg); ctx.set_knob("l1_waves", waves)
            for cap in (0, 3):   # a tiny resident grid: every wave strides over many items
                ctx.set_knob("g1_grid_per_cu", cap); ctx.set_knob("g2_grid_per_cu", cap)
                assert np.array_equal(ctx.msm_g1(pts, sc), want1), (wg, waves, cap)
                assert np.array_equal(ctx.msm_g2(p2, s2), want2), (wg, waves, cap)
    finally:
        _reset(ctx)


@pytest.mark.parametrize("limb29", [1, 2, 0])
def test_finisher_on_off_and_thresholds_agree_with_oracle(ctx, limb29):
    """finisher off = the item levels to the end (round 4's flow); on with the automatic threshold; on with thresholds that put it after
    level 1, 2, 3 (small item sizes make many levels); partial sums in the R' form (1), the standard form after a 29-bit level 1 (2),
    and 8 x 32-bit kernels throughout (0); G1 and G2"""
    n = (1 << 17) + 5
    pts, sc = _skewed(n, 9300)
    n2 = (1 << 15) + 7
    p2, s2 = _skewed(n2, 9400, g2=True)
    want1, want2 = cref.msm_g1(pts, sc), cref.msm_g2(p2, s2)
    try:
        assert ctx.lib.mi_debug_set_msm_limb29(ctx.h, limb29) == 0
        for plan in ((0, 0, 0, 0, 0), (11, 4, 2, 0, 0), (9, 16, 8, 0, 0)):
            assert ctx.lib.mi_debug_set_msm_plan(ctx.h, *plan) == 0
            for fin, fmax, fmin in ((0, 0, 2), (1, 0, 2), (1, 0, 0), (1, 17, 0), (1, 300, 1), (1, 1 << 20, 0)):   # fmin 0: the finisher right after level 1
                ctx.set_knob("finisher", fin); ctx.set_knob("finisher_max", fmax); ctx.set_knob("finisher_min_level", fmin)
                assert np.array_equal(ctx.msm_g1(pts, sc), want1), (limb29, plan, fin, fmax, fmin)
                assert np.array_equal(ctx.msm_g2(p2, s2), want2), (limb29, plan, fin, fmax, fmin)
    finally:
        _reset(ctx)


def test_finisher_all_pairs_one_point(ctx):
    """every pair the same point and scalar: every addition in every level and in the finisher's tree is a doubling (the special case
    of the R'-form and standard additions), G1 and G2"""
    n = 1 << 16
    same = cref.gen_g1(n, 9500); same[:] = same[0]
    ssc = cref.gen_scalars(n, 9501, 0); ssc[:] = ssc[0]
    n2 = 1 << 14
    same2 = cref.gen_g2(n2, 9502); same2[:] = same2[0]
    want1, want2 = cref.msm_g1(same, ssc), cref.msm_g2(same2, ssc[:n2])
    try:
        for plan in ((0, 0, 0, 0, 0), (5, 0, 0, 0, 0)):
            assert ctx.lib.mi_debug_set_msm_plan(ctx.h, *plan) == 0
            for fmax in (0, 40, 1 << 20):
                ctx.set_knob("finisher_max", fmax); ctx.set_knob("finisher_min_level", 0)
                assert np.array_equal(ctx.msm_g1(same, ssc), want1), (plan, fmax)
                assert np.array_equal(ctx.msm_g2(same2, ssc[:n2]), want2), (plan, fmax)
    finally:
        _reset(ctx)


def test_prove_with_round5_knobs_gives_the_oracle_bytes(ctx):
    """a whole proof (fixed-base tables and generic plans) under the combinations the benchmark may run with"""
    B = load_binding()
    log_n = 15
    N = 1 << log_n
    pk = synthetic_pk(log_n, N - 50, 300, 6161, n_committed=9)
    W = cref.gen_scalars(N - 50, 1, 1); a = cref.gen_scalars(N - 10, 2, 1); b = cref.gen_scalars(N - 10, 3, 0); c = cref.field_op(0, 2, a, b)
    r, s = cref.gen_scalars(2, 4, 0)
    want = cref.proof_write(cref.prove(pk, W, a, b, c, r, s)["raw"])
    try:
        for knob in ((0, 0, 0), (17, 18, 17)):
            assert ctx.lib.mi_debug_set_prove_fixed_base(ctx.h, *knob) == 0
            pkh = ctx.pk_load(pk)
            for cfg in ({}, {"l1_wg": 4}, {"l1_wg": 4, "l1_waves": 2, "g2_wg": 2}, {"finisher": 0}, {"finisher_max": 20, "l1_wg": 2, "z_waves": 2},
                        {"plain_scatter": 1, "count_per": 8}, {"l1_wg": 1}, {"finisher_min_level": 0, "finisher_max": 1 << 20}, {"z_count_fused": 0}):
                _reset(ctx)
                for k, v in cfg.items():
                    ctx.set_knob(k, v)
                got, _ = ctx.prove(pkh, W, a, b, c, r, s)
                assert B.proof_write(got["raw"]) == want, (knob, cfg)
            ctx.pk_free(pkh)
    finally:
        _reset(ctx)
        assert ctx.lib.mi_debug_set_prove_fixed_base(ctx.h, 0, 0, 0) == 0


@pytest.mark.parametrize("log_n", [13, 16, 18])
def test_z_digit_count_from_compute_h_equals_the_count_pass(ctx, log_n):
    """the Z MSM's digit count taken from computeH's last launch (mi_ctx::zhook) against the sort's own count pass and the oracle: c given
    and c = NULL (formed on the device), fewer constraints than the domain (zero padding), window widths 17 and 20 for Z"""
    B = load_binding()
    N = 1 << log_n
    nw, nc = N - 37, N - 5
    pk = synthetic_pk(log_n, nw, 40, 7000 + log_n)
    W = cref.gen_scalars(nw, 1, 1); a = cref.gen_scalars(nc, 2, 0); b = cref.gen_scalars(nc, 3, 0); c = cref.field_op(0, 2, a, b)
    r, s = cref.gen_scalars(2, 4, 0)
    want = cref.proof_write(cref.prove(pk, W, a, b, c, r, s)["raw"])
    try:
        for cz in (17, 20):
            assert ctx.lib.mi_debug_set_prove_fixed_base(ctx.h, 17, 17, cz) == 0
            pkh = ctx.pk_load(pk)
            for fused in (1, 0, 1):
                ctx.set_knob("z_count_fused", fused)
                before = ctx.counter("z_count_fused_launches")
                got, _ = ctx.prove(pkh, W, a, b, c, r, s)
                assert B.proof_write(got["raw"]) == want, (cz, fused, "c given")
                got, _ = ctx.prove(pkh, W, a, b, None, r, s)
                assert B.proof_write(got["raw"]) == want, (cz, fused, "c formed on the device")
                assert ctx.counter("z_count_fused_launches") == before + 2 * fused, (cz, fused)   # not vacuous: the fused launch ran, or did not
            ctx.pk_free(pkh)
    finally:
        _reset(ctx)
        assert ctx.lib.mi_debug_set_prove_fixed_base(ctx.h, 0, 0, 0) == 0


def test_z_digit_count_from_compute_h_with_two_slices_per_tile(ctx):
    """N = 2^24: the last launch's tile holds 2^10 elements = TWO slices of the Z sort.  Device-generated key and witness (as bench.py's); the
    proof with the fused count must equal the proof with the sort's own count pass (the path the oracle is compared with at every size)."""
    B = load_binding()
    log_n = 24
    N = 1 << log_n
    nw, npub, nc = N - 1000, 4097, N - 100
    rng = np.random.default_rng(77)
    inf_a = (rng.integers(0, 100, nw) < 10).astype(np.uint8); inf_b = (rng.integers(0, 100, nw) < 50).astype(np.uint8)
    na, nb, nk = int((inf_a == 0).sum()), int((inf_b == 0).sum()), nw - npub
    g1a, g1b, g1k, g1z, g2b = ctx.gen_g1(na, 1), ctx.gen_g1(nb, 2), ctx.gen_g1(nk, 3), ctx.gen_g1(N, 4), ctx.gen_g2(nb, 5)
    small = ctx.gen_g1(3, 6).download((3, 8)); small2 = ctx.gen_g2(2, 7).download((2, 16))
    pk = {"log_n": log_n, "nb_public": npub, "nb_wires": nw, "g1_a": (g1a.ptr, na), "g1_b": (g1b.ptr, nb), "g1_k": (g1k.ptr, nk), "g1_z": (g1z.ptr, N), "g2_b": (g2b.ptr, nb),
          "alpha1": small[0], "beta1": small[1], "delta1": small[2], "beta2": small2[0], "delta2": small2[1], "infinity_a": inf_a, "infinity_b": inf_b}
    pkh = ctx.pk_load(pk, device_points=True)
    W = ctx.gen_scalars(nw, 8, 1); a = ctx.gen_scalars(nc, 9, 1); b = ctx.gen_scalars(nc, 10, 0)
    c = ctx.alloc(32 * nc); ctx.field_op_dev(0, 2, c.ptr, a.ptr, b.ptr, nc)
    rs = ctx.gen_scalars(2, 11, 0).download((2, 4)); ctx.sync()
    try:
        got = {}
        before = ctx.counter("z_count_fused_launches")
        for fused in (0, 1, 0):
            ctx.set_knob("z_count_fused", fused)
            pr, _ = ctx.prove(pkh, W.ptr, a.ptr, b.ptr, c.ptr, rs[0], rs[1], device=True, n_wires=nw, n_constraints=nc)
            got.setdefault(fused, []).append(B.proof_write(pr["raw"]))
        assert got[0][0] == got[0][1] == got[1][0]
        assert ctx.counter("z_count_fused_launches") == before + 1   # the one proof with the knob on took its count from computeH
    finally:
        _reset(ctx)
        ctx.pk_free(pkh)
        for d in (g1a, g1b, g1k, g1z, g2b, W, a, b, c):
            d.free()
        assert ctx.lib.mi_ctx_trim(ctx.h) == 0


def test_flat_sort_item_size_agrees_with_oracle(ctx):
    """a proof whose FIVE MSMs all run over uniform scalars (flat sorts: ~200 entries in every bucket): the automatic level-1 item size of a
    flat sort (average / L2^k), the plan's own, and forced ones -- the same 164 bytes as the oracle's, and an MSM with a skewed mix beside it"""
    B = load_binding()
    log_n = 20
    N = 1 << log_n
    nw, nc = N - 37, N - 5
    pk = synthetic_pk(log_n, nw, 40, 7700)
    W = cref.gen_scalars(nw, 1, 0); a = cref.gen_scalars(nc, 2, 0); b = cref.gen_scalars(nc, 3, 0); c = cref.field_op(0, 2, a, b)
    r, s = cref.gen_scalars(2, 4, 0)
    want = cref.proof_write(cref.prove(pk, W, a, b, c, r, s)["raw"])
    n = (1 << 18) + 5
    pts, sc = _skewed(n, 9400)
    want_msm = cref.msm_g1(pts, sc)
    try:
        assert ctx.lib.mi_debug_set_prove_fixed_base(ctx.h, 17, 17, 17) == 0
        pkh = ctx.pk_load(pk)
        for flat in (0, 1, 20, 33, 0):
            ctx.set_knob("flat_item_l1", flat)
            got, _ = ctx.prove(pkh, W, a, b, c, r, s)
            assert B.proof_write(got["raw"]) == want, flat
            assert np.array_equal(ctx.msm_g1(pts, sc), want_msm), flat
        ctx.pk_free(pkh)
    finally:
        _reset(ctx)
        assert ctx.lib.mi_debug_set_prove_fixed_base(ctx.h, 0, 0, 0) == 0


def test_dense_sort_item_size_agrees_with_oracle(ctx):
    """the DENSE-sort rule (r6: a sort that is not flat in which >= half of the n x windows digits are entries sums level-1 items of 32): a proof
    whose wire values follow the census mix (74 % full-width: the rule triggers for A + K and B), through fixed-base tables -- automatic, off and
    forced sizes give the oracle's 164 bytes; a BASELINE-mix witness beside it (the rule must not trigger: ~0.3 of the digits) as well"""
    B = load_binding()
    log_n = 20
    N = 1 << log_n
    nw, nc = N - 37, N - 5
    pk = synthetic_pk(log_n, nw, 40, 7900)
    dist = B.dist_mix(23, 237, 0)
    cases = []
    for d in (dist, 1):
        W = cref.gen_scalars(nw, 11, d); a = cref.gen_scalars(nc, 12, d); b = cref.gen_scalars(nc, 13, 0); c = cref.field_op(0, 2, a, b)
        r, s = cref.gen_scalars(2, 14, 0)
        cases.append((W, a, b, c, r, s, cref.proof_write(cref.prove(pk, W, a, b, c, r, s)["raw"])))
    try:
        assert ctx.lib.mi_debug_set_prove_fixed_base(ctx.h, 19, 17, 20) == 0
        pkh = ctx.pk_load(pk)
        hits = {}
        for dense in (0, 1, 24, 48, 0):
            ctx.set_knob("dense_item_l1", dense)
            for k, (W, a, b, c, r, s, want) in enumerate(cases):
                before = ctx.counter("dense_item_sorts")
                got, st = ctx.prove(pkh, W, a, b, c, r, s)
                assert B.proof_write(got["raw"]) == want, (dense, k)
                hits[(dense, k)] = ctx.counter("dense_item_sorts") - before
        # census witness: A, K (one sort, two accumulations), B1, B2 take the rule's size -- and so does Z at THIS size (uniform h, but only 26
        # entries a bucket at N = 2^20 with 20-bit windows: below the flat rule's 64; at N = 2^23 Z is a flat sort with its own size).
        # BASELINE mix (~0.3 of the wire digits non-zero): Z alone.
        assert hits[(0, 0)] == 5 and hits[(24, 0)] == 5 and hits[(1, 0)] == 0, hits
        assert hits[(0, 1)] == 1 and hits[(48, 1)] == 1 and hits[(1, 1)] == 0, hits
        ctx.pk_free(pkh)
    finally:
        _reset(ctx)
        assert ctx.lib.mi_debug_set_prove_fixed_base(ctx.h, 0, 0, 0) == 0
