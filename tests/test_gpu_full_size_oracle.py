"""GPU parity at BASELINE.json configs[1] itself (FFT domain N = 2^23, nbPublic = 4097, infinity masks 10 % / 50 %,
WHIR scalar mix): the production plans -- 3-pass NTT with the composed twiddle tables, fixed-base sorts at c = 17..20,
the generic c = 16 sort with 2^19 keys -- are compared with the ORACLE (oracle/groth16_ref.c through cref), byte for
byte, at their real shape.  VERDICT r1 "next" item 1 / ADVICE r1 (tests/test_gpu_msm_prove.py:270).

One oracle prove at this size takes about 20 s on the GPU box's host cores; the module shares it between the tests.
The reference call whose bytes all of this stands for: groth16.Prove at /root/reference/mt.go:496.
"""
import time
import numpy as np
import pytest
import pyref as P
import cref
from helpers import *
from gpu_common import load_binding

pytestmark = pytest.mark.gpu
LOG_N = 23


@pytest.fixture(scope="module")
def ctx():
    B = load_binding()
    c = B.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def case(ctx):
    """Inputs are produced once by the device generators (bit-identical to cref.gen_*: asserted on a prefix here and
    in test_generators_match_oracle), downloaded, and handed to BOTH sides as host arrays."""
    N = 1 << LOG_N
    nb_wires, nb_public, n_constraints = N - 1000, 4097, N - 100
    seed = 0x57484952 + 1
    rng = np.random.default_rng(seed)
    inf_a = (rng.integers(0, 100, nb_wires) < 10).astype(np.uint8)
    inf_b = (rng.integers(0, 100, nb_wires) < 50).astype(np.uint8)
    na, nb, nk = int((inf_a == 0).sum()), int((inf_b == 0).sum()), nb_wires - nb_public

    def pull(d, shape):
        out = d.download(shape); d.free(); return out
    pk = {"log_n": LOG_N, "nb_public": nb_public, "nb_wires": nb_wires,
          "g1_a": pull(ctx.gen_g1(na, seed + 1), (na, 8)), "g1_b": pull(ctx.gen_g1(nb, seed + 2), (nb, 8)),
          "g1_k": pull(ctx.gen_g1(nk, seed + 3), (nk, 8)), "g1_z": pull(ctx.gen_g1(N, seed + 4), (N, 8)),
          "g2_b": pull(ctx.gen_g2(nb, seed + 5), (nb, 16)), "infinity_a": inf_a, "infinity_b": inf_b}
    small = cref.gen_g1(3, seed + 6); small2 = cref.gen_g2(2, seed + 7)
    pk.update(alpha1=small[0], beta1=small[1], delta1=small[2], beta2=small2[0], delta2=small2[1])
    assert np.array_equal(pk["g1_z"][:4096], cref.gen_g1(4096, seed + 4)) and np.array_equal(pk["g2_b"][:256], cref.gen_g2(256, seed + 5))
    assert cref.g1_on_curve(pk["g1_a"][-100000:]) and cref.g2_on_curve(pk["g2_b"][-20000:])
    W = pull(ctx.gen_scalars(nb_wires, seed + 8, 1), (nb_wires, 4))
    a = pull(ctx.gen_scalars(n_constraints, seed + 9, 1), (n_constraints, 4))
    b = pull(ctx.gen_scalars(n_constraints, seed + 10, 0), (n_constraints, 4))
    assert np.array_equal(W[:5000], cref.gen_scalars(5000, seed + 8, 1))
    c = cref.field_op(0, 2, a, b)
    r, s = cref.gen_scalars(2, seed + 11, 0)
    t0 = time.perf_counter()
    want, h = cref.prove(pk, W, a, b, c, r, s, want_h=True)
    print(f"oracle prove at N=2^{LOG_N}: {time.perf_counter() - t0:.1f} s on {cref.num_threads()} threads")
    return dict(pk=pk, W=W, a=a, b=b, c=c, r=r, s=s, want=want, h=h, n_constraints=n_constraints, nb=nb, inf_b=inf_b)


@pytest.mark.parametrize("knob,label", [((0, 0, 0), "fixed-base tables (automatic: c = 19 / 17 / 20)"), ((1, 1, 1), "generic c = 16 sort for every MSM")])
def test_prove_bytes_equal_oracle_at_baseline_size(ctx, case, knob, label):
    """mi_groth16_prove (host pointers in, the cgo path) -> proof bytes == oracle proof bytes, both MSM plans"""
    B = load_binding()
    assert ctx.lib.mi_debug_set_prove_fixed_base(ctx.h, *knob) == 0
    try:
        pkh = ctx.pk_load(case["pk"])
    finally:
        assert ctx.lib.mi_debug_set_prove_fixed_base(ctx.h, 0, 0, 0) == 0
    got, st = ctx.prove(pkh, case["W"], case["a"], case["b"], case["c"], case["r"], case["s"])
    ctx.pk_free(pkh)
    assert B.proof_write(got["raw"]) == cref.proof_write(case["want"]["raw"]), label
    assert len(B.proof_write(got["raw"])) == 164


def test_census_witness_at_baseline_size_equals_oracle(ctx, case):
    """The same key at N = 2^23 with the witness mix tools/wire_census.py derives from the reference's circuit (74 % of the wire values
    full-width, mtUtilities.go:494-532; MI_DIST_MIX through the device generator = cref's): 2.6 x the level-1 additions of the BASELINE mix
    through the production plan (fixed-base tables, c = 19 / 17 / 20), c formed on the device: proof bytes == oracle proof bytes"""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import wire_census
    B = load_binding()
    dist = B.dist_mix(*wire_census.census_mix_permille())
    pk, nc = case["pk"], case["n_constraints"]
    nw = pk["nb_wires"]
    W = ctx.gen_scalars(nw, 808, dist).download((nw, 4)); a = ctx.gen_scalars(nc, 809, dist).download((nc, 4))
    assert np.array_equal(W[:3000], cref.gen_scalars(3000, 808, dist))
    b = case["b"]
    c = cref.field_op(0, 2, a, b)
    t0 = time.perf_counter()
    want = cref.proof_write(cref.prove(pk, W, a, b, c, case["r"], case["s"])["raw"])
    print(f"oracle prove, census mix, N=2^{LOG_N}: {time.perf_counter() - t0:.1f} s")
    pkh = ctx.pk_load(pk)
    got, st = ctx.prove(pkh, W, a, b, None, case["r"], case["s"])
    ctx.pk_free(pkh)
    assert B.proof_write(got["raw"]) == want
    assert st["g1_level1_additions"] > 300e6, "the dense regime: > 300 M G1 level-1 additions a proof (BASELINE mix: ~200 M)"


def test_compute_h_equals_oracle_at_baseline_size(ctx, case):
    """all 2^23 coefficients of h (bit-reversed order, as gnark leaves them) against the oracle's"""
    got = ctx.compute_h(LOG_N, case["a"], case["b"], case["c"])
    assert np.array_equal(got, case["h"])


def test_full_length_msms_equal_oracle(ctx, case):
    """the Z MSM (N - 1 uniform h coefficients against pk.G1.Z) and the G2 MSM (filtered W against pk.G2.B), alone"""
    N = 1 << LOG_N
    got = ctx.msm_g1(case["pk"]["g1_z"][:N - 1], case["h"][:N - 1])
    assert np.array_equal(got, cref.msm_g1(case["pk"]["g1_z"][:N - 1], case["h"][:N - 1]))
    wb = np.ascontiguousarray(case["W"][case["inf_b"] == 0])
    got2 = ctx.msm_g2(case["pk"]["g2_b"], wb)
    assert np.array_equal(got2, cref.msm_g2(case["pk"]["g2_b"], wb))
    got1 = ctx.msm_g1(case["pk"]["g1_b"], wb)
    assert np.array_equal(got1, cref.msm_g1(case["pk"]["g1_b"], wb))


@pytest.mark.parametrize("mode", [0, 1])
def test_sharded_prove_two_ranks_at_baseline_size(case, mode):
    """N = 2^23 over a 2-rank group on device 0 (same partitioning, per-rank Pippenger with the automatic fixed-base tables, h slices
    and -- mode 1 -- bucket slices moved by the group's transport): proof bytes == the oracle's.  Ranks on
    DISTINCT devices have never run (this box has one GPU)."""
    B = load_binding()
    g = B.Group([0, 0])
    try:
        spk = g.pk_load(case["pk"])
        got, st = g.prove(spk, case["W"], case["a"], case["b"], case["c"], case["r"], case["s"], mode=mode)
        g.pk_free(spk)
        print(f"sharded prove N=2^{LOG_N}, 2 ranks on one device, mode {mode}, host inputs: {st['total_ms']:.1f} ms")
        assert B.proof_write(got["raw"]) == cref.proof_write(case["want"]["raw"])
    finally:
        g.close()
