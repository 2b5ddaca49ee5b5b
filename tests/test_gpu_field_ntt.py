"""GPU parity (through the C-ABI): device field / curve layer, generators, NTT, computeH."""
import os
import json
import numpy as np
import pytest
import pyref as P
import cref
from helpers import *
from gpu_common import load_binding

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def ctx():
    B = load_binding()
    c = B.Context(0)
    yield c
    c.close()


def test_field_ops_bit_exact(ctx):
    rng = P.SplitMix64(21)
    for field, mod, arr in ((0, P.R_MOD, fr_arr), (1, P.Q_MOD, fp_arr)):
        xs = [rng.fr() % mod for _ in range(500)] + [0, 1, mod - 1, mod - 2, 2, 0, mod - 1]
        ys = [rng.fr() % mod for _ in range(500)] + [0, mod - 1, mod - 1, 1, mod - 2, 5, 1]
        X, Y = arr(xs), arr(ys)
        for op in range(6):
            assert np.array_equal(ctx.field_op(field, op, X, Y), cref.field_op(field, op, X, Y)), (field, op)
    # large random sweep of the multiplier
    X = cref.gen_scalars(1 << 16, 1, 0); Y = cref.gen_scalars(1 << 16, 2, 0)
    assert np.array_equal(ctx.field_op(0, 2, X, Y), cref.field_op(0, 2, X, Y))
    assert np.array_equal(ctx.field_op(1, 2, X, Y), cref.field_op(1, 2, X, Y))


def test_montgomery_product_carry_edge_limbs(ctx):
    """Device multiplier (generated per-column asm, single and dual-product) on Montgomery residues with limbs of
    0xFFFFFFFF where a column's first product sits -- the case round 1's carry-less first product got wrong
    (field.cuh:259) -- against Python big integers."""
    from test_device_headers_on_host import adversarial_mont_pairs, raw_arr
    for field, mod in ((0, P.R_MOD), (1, P.Q_MOD)):
        xs, ys = adversarial_mont_pairs(mod, 77 + field, n_per=200)
        X, Y = raw_arr(xs), raw_arr(ys)
        ri = pow(1 << 256, -1, mod)
        want = {2: [x * y * ri % mod for x, y in zip(xs, ys)],
                6: [2 * x * y * ri % mod for x, y in zip(xs, ys)],
                7: [(x * y - y * y) * ri % mod for x, y in zip(xs, ys)],
                8: [x * x * ri % mod for x in xs]}
        for op, w in want.items():
            got = [cref.limbs_to_int(r) for r in ctx.field_op(field, op, X, Y)]
            assert got == w, (field, op, sum(a != b for a, b in zip(got, w)))


def test_generators_match_oracle(ctx):
    n = 3000
    B = load_binding()
    for dist in (0, 1, B.dist_mix(23, 237, 0), B.dist_mix(450, 250, 50), B.dist_mix(0, 0, 1000), B.dist_mix(1000, 0, 0), B.dist_mix(0, 0, 0)):
        assert np.array_equal(ctx.gen_scalars(n, 77, dist).download((n, 4)), cref.gen_scalars(n, 77, dist))
    # MI_DIST_MIX(bit, byte, u64): the shares are what the name says (per mille, the rest full-width)
    v = fr_vals(cref.gen_scalars(20000, 5, B.dist_mix(23, 237, 0)))
    share = lambda f: sum(1 for x in v if f(x)) / len(v)
    assert abs(share(lambda x: x < 2) - (0.023 + 0.237 * 2 / 256)) < 0.006 and abs(share(lambda x: x < 256) - 0.260) < 0.012 and share(lambda x: x >= 1 << 200) > 0.70
    assert np.array_equal(ctx.gen_g1(n, 5).download((n, 8)), cref.gen_g1(n, 5))
    assert np.array_equal(ctx.gen_g2(200, 6).download((200, 16)), cref.gen_g2(200, 6))


def test_curve_add_all_cases(ctx):
    g1 = cref.gen_g1(64, 5)
    neg = g1_arr([P.g1_neg(p) for p in g1_pts(g1[4:8])])
    a = np.concatenate([g1[:32], g1[:4], g1[4:8], np.zeros((2, 8), np.uint64), g1[8:9]])
    b = np.concatenate([g1[32:], g1[:4], neg, g1[9:10], np.zeros((1, 8), np.uint64), np.zeros((1, 8), np.uint64)])
    assert np.array_equal(ctx.ec_add(a, b), cref.g1_add(a, b))
    g2 = cref.gen_g2(24, 6)
    neg2 = g2_arr([P.g2_neg(p) for p in g2_pts(g2[2:4])])
    a2 = np.concatenate([g2[:8], g2[:2], g2[2:4], np.zeros((1, 16), np.uint64)])
    b2 = np.concatenate([g2[8:16], g2[:2], neg2, g2[3:4]])
    assert np.array_equal(ctx.ec_add(a2, b2, g2=True), cref.g2_add(a2, b2))


def test_ntt_golden_vectors(ctx):
    with open(os.path.join(GOLD, "ntt.json")) as f:
        g = json.load(f)
    for case in g["cases"]:
        a = [int(x, 16) for x in case["in"]]
        want = [int(x, 16) for x in case["out"]]
        assert fr_vals(ctx.ntt(fr_arr(a), case["log_n"], case["flags"])) == want, (case["log_n"], case["flags"])


@pytest.mark.parametrize("log_n", [0, 1, 2, 5, 8, 11, 12, 13, 16, 19, 20])
def test_ntt_all_modes_vs_oracle(ctx, log_n):
    n = 1 << log_n
    a = cref.gen_scalars(n, 100 + log_n, 0)
    modes = range(8) if log_n <= 16 else (1, 6, 3)
    for flags in modes:
        assert np.array_equal(ctx.ntt(a, log_n, flags), cref.ntt(a, log_n, flags)), flags


def test_ntt_small_tiles_multi_pass(ctx):
    """same kernels with the tile / radix knobs turned down: 3- and 4-pass plans at small n"""
    lib = ctx.lib
    try:
        for (log_e, mc, ms, log_n) in ((6, 3, 3, 10), (7, 5, 4, 13), (8, 8, 2, 14)):
            assert lib.mi_debug_set_ntt_plan(ctx.h, log_e, mc, ms) == 0
            a = cref.gen_scalars(1 << log_n, 7, 0)
            for flags in range(8):
                assert np.array_equal(ctx.ntt(a, log_n, flags), cref.ntt(a, log_n, flags)), (log_e, flags)
    finally:
        assert lib.mi_debug_set_ntt_plan(ctx.h, 9, 9, 7) == 0


def test_ntt_roundtrip_and_linearity_at_scale(ctx):
    """size-independent properties at a BASELINE-sized domain (2^23): inverse(forward(x)) == x,
    NTT(x + y) == NTT(x) + NTT(y), compared on device-downloaded samples."""
    log_n = 23
    n = 1 << log_n
    x = ctx.gen_scalars(n, 1, 0); y = ctx.gen_scalars(n, 2, 0)
    x0 = x.download((n, 4))
    ctx.ntt_dev(x.ptr, log_n, 0)            # forward DIF -> bit-reversed
    ctx.ntt_dev(x.ptr, log_n, 1 | 4)        # inverse DIT (bit-reversed in) -> natural
    assert np.array_equal(x.download((n, 4)), x0)
    # coset DIT reads x as bit-reversed coefficients and yields natural-order coset evaluations;
    # the inverse coset DIF maps those back to bit-reversed coefficients: the same array again
    ctx.ntt_dev(x.ptr, log_n, 2 | 4)
    assert not np.array_equal(x.download((4, 4)), x0[:4])
    ctx.ntt_dev(x.ptr, log_n, 1 | 2)
    assert np.array_equal(x.download((n, 4)), x0)
    # linearity on the device: NTT(x) + NTT(y) == NTT(x + y)
    s = ctx.alloc(32 * n)
    ctx.field_op_dev(0, 0, s.ptr, x.ptr, y.ptr, n)
    for d in (x, y, s):
        ctx.ntt_dev(d.ptr, log_n, 0)
    ctx.field_op_dev(0, 0, x.ptr, x.ptr, y.ptr, n)
    assert np.array_equal(x.download((n, 4)), s.download((n, 4)))
    s.free()
    for d in (x, y):
        d.free()


@pytest.mark.parametrize("nc", [1, 7, 1000, 5000, 70000])
def test_compute_h_vs_oracle(ctx, nc):
    log_n = max(nc - 1, 0).bit_length()
    a = cref.gen_scalars(nc, 1, 1); b = cref.gen_scalars(nc, 2, 0); c = cref.field_op(0, 2, a, b)
    assert np.array_equal(ctx.compute_h(log_n, a, b, c), cref.compute_h(log_n, a, b, c))


def test_compute_h_golden_toy1000(ctx):
    z = np.load(os.path.join(GOLD, "prove_toy1000.npz"))
    assert np.array_equal(ctx.compute_h(int(z["log_n"]), z["a"], z["b"], z["c"]), z["h"])


def test_context_lifecycle_and_two_contexts():
    """init / shutdown repeatedly (handles, streams, pinned buffers are released), two live contexts side by side"""
    B = load_binding()
    a = cref.gen_scalars(1 << 10, 5, 0)
    want = cref.ntt(a, 10, 1)
    for _ in range(10):
        c = B.Context(0)
        assert np.array_equal(c.ntt(a, 10, 1), want)
        c.close()
    c1, c2 = B.Context(0), B.Context(0)
    pts = cref.gen_g1(500, 1); sc = cref.gen_scalars(500, 2, 1)
    r1, r2 = c1.msm_g1(pts, sc), c2.msm_g1(pts, sc)
    assert np.array_equal(r1, r2) and np.array_equal(r1, cref.msm_g1(pts, sc))
    c1.close(); c2.close()


def test_invalid_arguments_are_rejected_not_executed(ctx):
    B = load_binding()
    with pytest.raises(B.MiError):
        ctx.ntt(np.zeros((4, 4), np.uint64), 29, 0)            # log_n > 28
    with pytest.raises(B.MiError):
        ctx.ntt(np.zeros((4, 4), np.uint64), 2, 8)             # unknown flag bit
    with pytest.raises(B.MiError):
        ctx.compute_h(2, np.zeros((5, 4), np.uint64), np.zeros((5, 4), np.uint64), np.zeros((5, 4), np.uint64))   # more constraints than the domain
    assert ctx.lib.mi_debug_set_msm_plan(ctx.h, 1, 0, 0, 0, 0) != 0 and ctx.lib.mi_debug_set_msm_plan(ctx.h, 17, 0, 0, 0, 0) != 0
    assert ctx.lib.mi_debug_set_msm_plan(ctx.h, 8, 1, 8, 0, 0) != 0      # items of one entry never converge
    assert ctx.lib.mi_debug_set_ntt_plan(ctx.h, 13, 9, 7) != 0
    assert b"" == b"" and ctx.lib.mi_last_error(ctx.h) is not None


@pytest.mark.parametrize("log_n", [7, 9, 13, 16, 18])
def test_ntt_register_and_lds_stage_paths_agree_with_oracle(ctx, log_n):
    """every mode with the wavefront-register stages (default for radices >= 2^7) and with every stage through LDS; computeH with
    and without its data-layout twiddle / coset tables"""
    n = 1 << log_n
    a = cref.gen_scalars(n, 400 + log_n, 0)
    want = {f: cref.ntt(a, log_n, f) for f in range(8)}
    b = cref.gen_scalars(n, 401 + log_n, 1); c = cref.field_op(0, 2, a, b)
    want_h = cref.compute_h(log_n, a[: n - 3], b[: n - 3], c[: n - 3])
    try:
        for on, dmin in ((1, 7), (0, 29), (1, 29), (0, 7)):
            assert ctx.lib.mi_debug_set_ntt_wave_stages(ctx.h, on, dmin) == 0
            for f in range(8):
                assert np.array_equal(ctx.ntt(a, log_n, f), want[f]), (on, f)
            assert np.array_equal(ctx.compute_h(log_n, a[: n - 3], b[: n - 3], c[: n - 3]), want_h), (on, dmin)
    finally:
        assert ctx.lib.mi_debug_set_ntt_wave_stages(ctx.h, 1, 12) == 0


@pytest.mark.parametrize("log_n,plan", [(10, None), (14, (8, 7, 7)), (16, None), (17, (10, 10, 7)), (20, None), (21, (9, 7, 7)), (16, (10, 8, 8)),
                                         (17, (9, 9, 8)), (19, (10, 3, 8))])
def test_compute_h_with_and_without_the_fused_launches(ctx, log_n, plan):
    """computeH with its two fused launches -- the inverse transform's last pass + the coset transform's first pass of a and b
    (bit 0; wherever the plan's contiguous radix is >= 2^7), and the coset transform's last pass of a and of b + the product + the last
    transform's first pass (bit 1; wherever the plan's first radix is 2^7 -- tiles of 2, 4 and 8 sub-blocks, two- and three-pass plans
    here -- or 2^8 -- the plans of N = 2^24 and 2^26: tiles of 4 and 2 columns, with and without a contiguous pair) -- in every combination, against the oracle: all of h, with zero padding, c unrelated to a b"""
    n = 1 << log_n
    nc = n - 11
    a = cref.gen_scalars(nc, 600 + log_n, 1); b = cref.gen_scalars(nc, 601 + log_n, 0); c = cref.gen_scalars(nc, 602 + log_n, 0)
    want = cref.compute_h(log_n, a, b, c)
    try:
        if plan:
            assert ctx.lib.mi_debug_set_ntt_plan(ctx.h, *plan) == 0
        for on in (7, 0, 1, 2, 3, 4, 5, 6, 7):   # bit 2 (r4): c's last pass + the last transform's last pass, which subtracts it, as one launch
            assert ctx.lib.mi_debug_set_ntt_fuse_pair(ctx.h, on) == 0
            assert np.array_equal(ctx.compute_h(log_n, a, b, c), want), (on, log_n)
            if on in (7, 4):   # c = NULL: formed on the device as a o b, against the oracle given that c
                assert np.array_equal(ctx.compute_h(log_n, a, b, None), cref.compute_h(log_n, a, b, cref.field_op(0, 2, a, b))), (on, log_n, "derived c")
    finally:
        assert ctx.lib.mi_debug_set_ntt_fuse_pair(ctx.h, 7) == 0
        if plan:
            assert ctx.lib.mi_debug_set_ntt_plan(ctx.h, 9, 9, 7) == 0
