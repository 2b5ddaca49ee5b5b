"""CPU suite: randomized (hypothesis) properties of the oracle's C restatement and of the device arithmetic headers
compiled for the host, against python big integers."""
import ctypes as C
import os
import subprocess
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st
import pyref as P
import cref
from helpers import *

HERE = os.path.dirname(os.path.abspath(__file__))
fr_el = st.integers(min_value=0, max_value=P.R_MOD - 1)
fp_el = st.integers(min_value=0, max_value=P.Q_MOD - 1)
edge = st.sampled_from([0, 1, 2, P.R_MOD - 1, P.R_MOD - 2, (1 << 253), (1 << 128) - 1, (1 << 64), (1 << 32) - 1])


@pytest.fixture(scope="module")
def emu():
    so = os.path.join(HERE, "emu", "libemu.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-DMI_CHECK_NOWRAP", "-shared", "-fPIC", "-o", so, os.path.join(HERE, "emu", "emu.cpp")])
    return C.CDLL(so)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


@settings(max_examples=60, deadline=None)
@given(st.lists(st.tuples(st.one_of(fr_el, edge), st.one_of(fr_el, edge)), min_size=1, max_size=8))
def test_fr_ring_axioms_in_both_implementations(emu, pairs):
    xs = [a % P.R_MOD for a, _ in pairs]; ys = [b % P.R_MOD for _, b in pairs]
    X, Y = fr_arr(xs), fr_arr(ys)
    for op, f in ((0, lambda a, b: (a + b) % P.R_MOD), (1, lambda a, b: (a - b) % P.R_MOD), (2, lambda a, b: a * b % P.R_MOD)):
        want = [f(a, b) for a, b in zip(xs, ys)]
        assert fr_vals(cref.field_op(0, op, X, Y)) == want
        Z = np.zeros_like(X)
        emu.emu_field_op(0, op, _p(Z), _p(X), _p(Y), C.c_size_t(len(xs)))
        assert fr_vals(Z) == want


@settings(max_examples=30, deadline=None)
@given(st.lists(fp_el, min_size=1, max_size=6))
def test_fp_inverse_is_inverse(emu, xs):
    X = fp_arr(xs)
    inv = cref.field_op(1, 3, X)
    prod = fp_vals(cref.field_op(1, 2, X, inv))
    assert prod == [0 if x == 0 else 1 for x in xs]
    Z = np.zeros_like(X)
    emu.emu_field_op(1, 3, _p(Z), _p(X), _p(X), C.c_size_t(len(xs)))
    assert np.array_equal(Z, inv)


@settings(max_examples=15, deadline=None)
@given(st.integers(min_value=1, max_value=40), st.integers(min_value=0, max_value=2**32), st.sampled_from([0, 1]),
       st.integers(min_value=2, max_value=9), st.integers(min_value=1, max_value=4), st.integers(min_value=2, max_value=6))
def test_msm_pipeline_random_shapes(emu, n, seed, dist, c, G, L):
    """the kernel bodies of msm_core.cuh under random (n, window bits, slices, item size) against the oracle"""
    pts = cref.gen_g1(n, seed); sc = cref.gen_scalars(n, seed + 1, dist)
    out = np.zeros(8, np.uint64)
    emu.emu_msm_g1(_p(out), _p(pts), _p(sc), n, 1, c, G, L, 2, 5)
    want = cref.msm_g1(pts, sc)
    assert np.array_equal(out, want[:8]) or (not want[8:].any() and not out.any())


@settings(max_examples=12, deadline=None)
@given(st.integers(min_value=1, max_value=9), st.integers(min_value=0, max_value=7), st.integers(min_value=0, max_value=2**32))
def test_ntt_pass_plans_random(emu, log_n, flags, seed):
    a = cref.gen_scalars(1 << log_n, seed, 0)
    for (log_e, mc, ms) in ((3, 3, 2), (5, 4, 3), (9, 9, 7)):
        got = a.copy()
        assert emu.emu_ntt(_p(got), log_n, flags, log_e, mc, ms, 32, 1 << log_n) >= 1
        assert np.array_equal(got, cref.ntt(a, log_n, flags)), (log_e, flags)


@settings(max_examples=25, deadline=None)
@given(st.integers(min_value=0, max_value=P.R_MOD - 1), st.integers(min_value=0, max_value=2**32))
def test_scalar_mul_distributes_over_msm(k, seed):
    """MSM(P, k*s) == k * MSM(P, s) on the oracle (what the blinding relations of the proof rely on)"""
    n = 7
    pts = cref.gen_g1(n, seed); sc = cref.gen_scalars(n, seed + 3, 1)
    ks = fr_arr([(k * v) % P.R_MOD for v in fr_vals(sc)])
    lhs = cref.msm_g1(pts, ks)
    base = cref.msm_g1(pts, sc)
    if not base[8:].any():
        assert not lhs[8:].any()
    else:
        assert np.array_equal(lhs[:8], cref.g1_scalar_mul(base[:8], k)) or (k == 0 and not lhs[8:].any())
