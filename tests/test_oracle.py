"""CPU suite: pins the oracle itself.  (a) pyref against independent definitions,
(b) the C restatement against pyref, (c) both against the committed golden fixtures."""
import json
import os
import numpy as np
import pytest
import pyref as P
import cref
from helpers import *

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_constants_rederived():
    # SURVEY 8a a4/a11 "VERIFIED" constants, re-derived here with python integers
    x = 4965661367192848881
    assert P.Q_MOD == 36 * x**4 + 36 * x**3 + 24 * x**2 + 6 * x + 1
    assert P.R_MOD == 36 * x**4 + 36 * x**3 + 18 * x**2 + 6 * x + 1
    assert pow(P.FR_ROOT_2_28, 1 << 28, P.R_MOD) == 1 and pow(P.FR_ROOT_2_28, 1 << 27, P.R_MOD) == P.R_MOD - 1
    assert (P.R_MOD - 1) % (1 << 28) == 0 and (P.R_MOD - 1) % (1 << 29) != 0
    # 5 generates Fr*: 5^((r-1)/p) != 1 for the prime factors of r-1
    for p in (2, 3, 13, 29, 983, 11003, 237073, 405928799, 1670836401704629, 13818364434197438864469338081):
        assert (P.R_MOD - 1) % p == 0 and pow(5, (P.R_MOD - 1) // p, P.R_MOD) != 1
    assert P.Q_MOD % 4 == 3
    assert P.g1_is_on_curve(P.G1_GEN) and P.g2_is_on_curve(P.G2_GEN)
    assert P.g1_mul(P.G1_GEN, P.R_MOD) is None and P.g2_mul(P.G2_GEN, P.R_MOD) is None


def test_pyref_ntt_against_dft_definition():
    rng = P.SplitMix64(7)
    for logn in (1, 3, 5):
        n = 1 << logn
        dom = P.Domain(n)
        a = [rng.fr() for _ in range(n)]
        nat = P.dft_definition(a, dom.gen)
        assert P.bit_reverse_perm(P.fft(dom, a, P.DIF)) == nat
        assert P.fft(dom, P.bit_reverse_perm(a), P.DIT) == nat
        cos = [sum(a[i] * pow(5 * pow(dom.gen, k, P.R_MOD), i, P.R_MOD) for i in range(n)) % P.R_MOD for k in range(n)]
        assert P.fft(dom, P.bit_reverse_perm(a), P.DIT, coset=True) == cos
        assert P.bit_reverse_perm(P.fft(dom, a, P.DIF, coset=True)) == cos
        assert P.fft_inverse(dom, cos, P.DIF, coset=True) == P.bit_reverse_perm(a)
        assert P.fft_inverse(dom, P.bit_reverse_perm(cos), P.DIT, coset=True) == a
        assert P.fft_inverse(dom, nat, P.DIF) == P.bit_reverse_perm(a)


def test_pyref_compute_h_is_the_quotient():
    cs = P.ToyR1CS(11, 3, 5)
    w, a, b, c = cs.solve()
    dom = P.Domain(cs.nb_constraints)
    h = P.bit_reverse_perm(P.compute_h(a, b, c, dom))
    assert h[dom.n - 1] == 0
    # a(x)b(x) - c(x) == h(x) (x^n - 1) at a random x, with a,b,c interpolated over <w>
    x = 0x1234567
    L = P.lagrange_at(dom, x)
    ev = lambda v: sum(vi * Li for vi, Li in zip(v + [0] * (dom.n - len(v)), L)) % P.R_MOD
    hx = sum(hi * pow(x, i, P.R_MOD) for i, hi in enumerate(h)) % P.R_MOD
    assert (ev(a) * ev(b) - ev(c) - hx * (pow(x, dom.n, P.R_MOD) - 1)) % P.R_MOD == 0


@pytest.mark.parametrize("nc,npub,seed", [(13, 3, 42), (30, 5, 43)])
def test_pyref_toy_proof_satisfies_groth16_equation(nc, npub, seed):
    cs = P.ToyR1CS(nc, npub, seed); td = P.ToyTrapdoor(seed)
    pk, exps, dom = P.toy_setup(cs, td)
    rng = P.SplitMix64(seed + 1); r, s = rng.fr(), rng.fr()
    pr = P.toy_prove(cs, pk, dom, r, s)
    assert P.trapdoor_check(cs, td, exps, pr, r, s)
    bad = dict(pr); bad["krs"] = P.g1_add(pr["krs"], P.G1_GEN)
    assert not P.trapdoor_check(cs, td, exps, bad, r, s)


def test_pyref_pippenger_equals_definition():
    rng = P.SplitMix64(3)
    pts = [P.synth_g1_point(rng) for _ in range(24)] + [None]
    sc = [P.synth_scalar(rng, "whir") for _ in range(23)] + [P.R_MOD - 1, 5]
    assert P.msm_naive(P.F1, pts, sc) == P.msm_pippenger(P.F1, pts, sc, 5)
    g2p = [P.g2_mul(P.G2_GEN, 3 + i) for i in range(6)]
    assert P.msm_naive(P.F2, g2p, sc[:6]) == P.msm_pippenger(P.F2, g2p, sc[:6], 4)


# ------------------------------------------------------------------ C restatement vs pyref
def test_c_field_ops_match_python():
    rng = P.SplitMix64(11)
    for field, mod, arr, vals in ((0, P.R_MOD, fr_arr, fr_vals), (1, P.Q_MOD, fp_arr, fp_vals)):
        xs = [rng.fr() % mod for _ in range(40)] + [0, 1, mod - 1, mod - 2, 2]
        ys = [rng.fr() % mod for _ in range(40)] + [0, mod - 1, mod - 1, 1, mod - 2]
        X, Y = arr(xs), arr(ys)
        assert vals(cref.field_op(field, 0, X, Y)) == [(x + y) % mod for x, y in zip(xs, ys)]
        assert vals(cref.field_op(field, 1, X, Y)) == [(x - y) % mod for x, y in zip(xs, ys)]
        assert vals(cref.field_op(field, 2, X, Y)) == [(x * y) % mod for x, y in zip(xs, ys)]
        assert vals(cref.field_op(field, 3, X)) == [pow(x, mod - 2, mod) for x in xs]
        # to_mont/from_mont round trip on raw canonical limbs
        raw = np.array([cref.int_to_limbs(x) for x in xs], dtype=np.uint64)
        assert np.array_equal(cref.field_op(field, 4, raw), X)
        assert np.array_equal(cref.field_op(field, 5, X), raw)


def test_c_generators_are_valid_and_curve_add_matches_python():
    g1 = cref.gen_g1(40, 1); g2 = cref.gen_g2(12, 2)
    assert cref.g1_on_curve(g1) and cref.g2_on_curve(g2)
    p1, p2 = g1_pts(g1), g2_pts(g2)
    assert all(P.g1_is_on_curve(p) for p in p1) and all(P.g2_is_on_curve(p) for p in p2)
    assert all(P.g2_mul(p, P.R_MOD) is None for p in p2[:3])   # r-torsion
    # add: generic, doubling, inverse, infinity operands
    a = p1[:8] + [p1[0], p1[1], None, p1[2], None]
    b = p1[8:16] + [p1[0], P.g1_neg(p1[1]), p1[3], None, None]
    assert g1_pts(cref.g1_add(g1_arr(a), g1_arr(b))) == [P.g1_add(x, y) for x, y in zip(a, b)]
    a2 = p2[:4] + [p2[0], p2[1], None]
    b2 = p2[4:8] + [p2[0], P.g2_neg(p2[1]), p2[2]]
    assert g2_pts(cref.g2_add(g2_arr(a2), g2_arr(b2))) == [P.g2_add(x, y) for x, y in zip(a2, b2)]


@pytest.mark.parametrize("logn", [1, 3, 6])
def test_c_ntt_all_modes_match_python(logn):
    n = 1 << logn
    rng = P.SplitMix64(logn)
    a = [rng.fr() for _ in range(n)]
    dom = P.Domain(n)
    A = fr_arr(a)
    for inverse in (0, 1):
        for coset in (0, 1):
            for dit in (0, 1):
                flags = inverse * 1 | coset * 2 | dit * 4
                f = P.fft_inverse if inverse else P.fft
                want = f(dom, a, P.DIT if dit else P.DIF, coset=bool(coset))
                assert fr_vals(cref.ntt(A, logn, flags)) == want, flags


def test_c_compute_h_matches_python():
    cs = P.ToyR1CS(50, 4, 9)
    w, a, b, c = cs.solve()
    dom = P.Domain(cs.nb_constraints)
    assert fr_vals(cref.compute_h(dom.log_n, fr_arr(a), fr_arr(b), fr_arr(c))) == P.compute_h(a, b, c, dom)


@pytest.mark.parametrize("n,dist", [(1, 0), (2, 1), (37, 0), (255, 1)])
def test_c_msm_matches_definition(n, dist):
    pts = cref.gen_g1(n, 77 + n); sc = cref.gen_scalars(n, 5 + n, dist)
    if n > 4:
        pts[3] = 0                       # a point at infinity
        sc[1] = fr_arr([P.R_MOD - 1])[0]  # extreme scalars
        sc[2] = 0
    want = P.msm_pippenger(P.F1, g1_pts(pts), fr_vals(sc), 6) if n > 40 else P.msm_naive(P.F1, g1_pts(pts), fr_vals(sc))
    assert g1_from_jac(cref.msm_g1(pts, sc)) == want
    assert g1_from_jac(cref.msm_g1(pts, sc, naive=True)) == want
    if n <= 37:
        p2 = cref.gen_g2(n, 99 + n)
        want2 = P.msm_naive(P.F2, g2_pts(p2), fr_vals(sc))
        assert g2_from_jac(cref.msm_g2(p2, sc)) == want2


def test_c_msm_canonical_flag_and_window_sizes():
    n = 3000  # c = 9 path
    pts = cref.gen_g1(n, 1234); sc = cref.gen_scalars(n, 4321, 1)
    got = cref.msm_g1(pts, sc)
    canon = cref.field_op(0, 5, sc)
    assert np.array_equal(cref.msm_g1(pts, canon, flags=1), got)
    # linearity: MSM(P, s) + MSM(P, t) == MSM(P, s+t)
    t = cref.gen_scalars(n, 999, 0)
    st = cref.field_op(0, 0, sc, t)
    lhs = cref.g1_sum(np.stack([got, cref.msm_g1(pts, t)]))
    assert np.array_equal(lhs, cref.msm_g1(pts, st))


def test_c_prove_matches_python_and_trapdoor():
    cs = P.ToyR1CS(21, 3, 77); td = P.ToyTrapdoor(77)
    pk, exps, dom = P.toy_setup(cs, td)
    rng = P.SplitMix64(5); r, s = rng.fr(), rng.fr()
    want = P.toy_prove(cs, pk, dom, r, s)
    w, a, b, c = cs.solve()
    got, h = cref.prove(toy_pk_arrays(pk), fr_arr(w), fr_arr(a), fr_arr(b), fr_arr(c), fr_arr([r])[0], fr_arr([s])[0], want_h=True)
    assert g1_pts(got["ar"]) == [want["ar"]] and g1_pts(got["krs"]) == [want["krs"]] and g2_pts(got["bs"]) == [want["bs"]]
    assert fr_vals(h) == want["h"]
    assert cref.proof_write(got["raw"]) == P.proof_bytes(want)
    chk = {"ar": g1_pts(got["ar"])[0], "bs": g2_pts(got["bs"])[0], "krs": g1_pts(got["krs"])[0], "h": fr_vals(h)}
    assert P.trapdoor_check(cs, td, exps, chk, r, s)


def test_encoding_known_answers():
    # rule of SURVEY 8a a12: generator (1,2): y=2 <= (q-1)/2 -> flag 0b10
    assert P.g1_compress(P.G1_GEN).hex() == "80" + "00" * 30 + "01"
    assert P.g1_compress(P.g1_neg(P.G1_GEN)).hex() == "c0" + "00" * 30 + "01"
    assert P.g1_compress(None).hex() == "40" + "00" * 31
    assert cref.g1_compress(g1_arr([P.G1_GEN])[0]) == P.g1_compress(P.G1_GEN)
    assert cref.g1_compress(g1_arr([None])[0]) == P.g1_compress(None)
    g2 = g2_arr([P.G2_GEN, P.g2_neg(P.G2_GEN), None])
    for row, pt in zip(g2, (P.G2_GEN, P.g2_neg(P.G2_GEN), None)):
        assert cref.g2_compress(row) == P.g2_compress(pt)


# ------------------------------------------------------------------ golden fixtures
def _golden(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)


def test_golden_fixtures_reproduce():
    """tests/golden/*.json were written by oracle/gen_golden.py from pyref; both oracles must
    still reproduce them (guards the oracle against drift)."""
    g = _golden("ntt.json")
    for case in g["cases"]:
        a = [int(x, 16) for x in case["in"]]
        want = [int(x, 16) for x in case["out"]]
        assert fr_vals(cref.ntt(fr_arr(a), case["log_n"], case["flags"])) == want
    g = _golden("msm.json")
    for case in g["g1"]:
        pts = [None if p is None else (int(p[0], 16), int(p[1], 16)) for p in case["points"]]
        sc = [int(x, 16) for x in case["scalars"]]
        want = None if case["out"] is None else (int(case["out"][0], 16), int(case["out"][1], 16))
        assert g1_from_jac(cref.msm_g1(g1_arr(pts), fr_arr(sc))) == want
    g = _golden("prove.json")
    cs = P.ToyR1CS(g["nb_constraints"], g["nb_public"], g["seed"]); td = P.ToyTrapdoor(g["seed"])
    pk, exps, dom = P.toy_setup(cs, td)
    w, a, b, c = cs.solve()
    r, s = int(g["r"], 16), int(g["s"], 16)
    got = cref.prove(toy_pk_arrays(pk), fr_arr(w), fr_arr(a), fr_arr(b), fr_arr(c), fr_arr([r])[0], fr_arr([s])[0])
    assert cref.proof_write(got["raw"]).hex() == g["proof_bytes"]


def test_golden_npz_fixtures_reproduce_with_c_oracle():
    for name in ("msm_g1_4096_uniform.npz", "msm_g1_4096_whir.npz"):
        z = np.load(os.path.join(GOLD, name))
        got = cref.msm_g1(z["points"], z["scalars"])
        assert np.array_equal(got[:8], z["out"])
    z = np.load(os.path.join(GOLD, "msm_g2_512_whir.npz"))
    assert np.array_equal(cref.msm_g2(z["points"], z["scalars"])[:16], z["out"])
    z = np.load(os.path.join(GOLD, "prove_toy1000.npz"))
    pk = {k: z[k] for k in ("g1_a", "g1_b", "g1_k", "g1_z", "g2_b", "alpha1", "beta1", "delta1", "beta2", "delta2", "infinity_a", "infinity_b")}
    pk.update(log_n=int(z["log_n"]), nb_public=int(z["nb_public"]), nb_wires=int(z["nb_wires"]))
    got, h = cref.prove(pk, z["W"], z["a"], z["b"], z["c"], z["r"], z["s"], want_h=True)
    assert np.array_equal(h, z["h"])
    assert np.array_equal(got["ar"], z["ar"]) and np.array_equal(got["bs"], z["bs"]) and np.array_equal(got["krs"], z["krs"])
    assert cref.proof_write(got["raw"]) == bytes(z["proof_bytes"])


def test_pedersen_commit_and_fold_match_python():
    """BSB22 commitment = MultiExp over the Pedersen basis; Fold = sum challenge^i * P_i"""
    basis = cref.gen_g1(50, 71); vals = cref.gen_scalars(37, 72, 1)
    assert g1_pts(cref.pedersen_msm(basis, vals)) == [P.pedersen_commit(g1_pts(basis), fr_vals(vals))]
    pts = cref.gen_g1(3, 73); ch = cref.gen_scalars(1, 74, 0)[0]
    assert g1_pts(cref.pedersen_fold(pts, ch)) == [P.pedersen_fold(g1_pts(pts), fr_vals(ch)[0])]


def test_batch_scalar_mul_matches_python():
    """BatchScalarMultiplicationG1/G2 (the bulk of groth16.Setup)"""
    sc = cref.gen_scalars(12, 91, 1); sc[0] = 0; sc[1] = fr_arr([1])[0]; sc[2] = fr_arr([P.R_MOD - 1])[0]
    out = cref.batch_scalar_mul(g1_arr([P.G1_GEN])[0], sc)
    assert g1_pts(out) == [P.g1_mul(P.G1_GEN, v) for v in fr_vals(sc)]
    out2 = cref.batch_scalar_mul(g2_arr([P.G2_GEN])[0], sc[:5], g2=True)
    assert g2_pts(out2) == [P.g2_mul(P.G2_GEN, v) for v in fr_vals(sc[:5])]
