"""GPU parity ABOVE the benchmark size, against the ORACLE (oracle/groth16_ref.c through cref), at the sizes where the
library switches plans (VERDICT r2 "next" item 1; BASELINE.json configs[2], FFT domain N = 2^26):

  computeH   2^24 (2^10-element tiles, radices 8/8/8), 2^25 and 2^26 (10-bit contiguous pass; 8 GB of direct twiddle /
             coset tables at 2^26) -- every coefficient of h
  NTT        all 8 flag combinations at 2^20; the DIT / coset modes at 2^24
  G1 MSM     2^26 uniform pairs on arbitrary bases: the generic c = 16 path through the LDS-staged two-pass sort with
             2^30 entries (what N = 2^26 uses for A+K and B)
  G2 MSM     2^25 pairs, WHIR scalar mix (the B2 MSM of an N = 2^26 key)

Inputs come from the device generators (bit-identical to cref.gen_*: asserted on a prefix), are downloaded once and handed
to the oracle as host arrays.  The oracle side takes 10-60 s per case on the GPU box's host cores.
The reference call whose results all of this stands for: groth16.Prove at /root/reference/mt.go:496.
"""
import time
import numpy as np
import pytest
import cref
from gpu_common import load_binding

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    B = load_binding()
    c = B.Context(0)
    yield c
    c.close()


def _timed(label, f):
    t0 = time.perf_counter()
    out = f()
    print(f"{label}: {time.perf_counter() - t0:.1f} s")
    return out


@pytest.mark.parametrize("log_n", [24, 25, 26])
def test_compute_h_equals_oracle_above_baseline_size(ctx, log_n):
    """all 2^log_n coefficients of h; n_constraints < N exercises the fused zero padding; c is NOT a*b on the last rows
    (the six-transform identity holds for any c)"""
    N = 1 << log_n
    n_constraints = N - 100
    seed = 0x57484952 + 2600 + log_n
    a = ctx.gen_scalars(n_constraints, seed, 1); b = ctx.gen_scalars(n_constraints, seed + 1, 0); c = ctx.alloc(32 * n_constraints)
    ctx.field_op_dev(0, 2, c.ptr, a.ptr, b.ptr, n_constraints - 1000)
    ctx.field_op_dev(0, 0, c.ptr + 32 * (n_constraints - 1000), a.ptr, b.ptr, 1000)   # last 1000 rows: c = a + b
    h = ctx.alloc(32 * N)
    ctx.compute_h_dev(log_n, a.ptr, b.ptr, c.ptr, n_constraints, h.ptr)
    ha, hb, hc = a.download((n_constraints, 4)), b.download((n_constraints, 4)), c.download((n_constraints, 4))
    got = h.download((N, 4))
    for d in (a, b, c, h):
        d.free()
    assert np.array_equal(ha[:3000], cref.gen_scalars(3000, seed, 1))
    want = _timed(f"oracle computeH at N=2^{log_n}", lambda: cref.compute_h(log_n, ha, hb, hc))
    assert np.array_equal(got, want)


@pytest.mark.parametrize("flags", range(8))
def test_ntt_all_modes_at_2p20(ctx, flags):
    a = cref.gen_scalars(1 << 20, 7000 + flags, 0)
    assert np.array_equal(ctx.ntt(a, 20, flags), cref.ntt(a, 20, flags))


@pytest.mark.parametrize("flags", [2 | 4, 1 | 2 | 4, 1 | 2])
def test_ntt_coset_modes_at_2p24(ctx, flags):
    """MI_NTT_COSET | MI_NTT_DIT (what computeH's forward transforms are), the inverse of it, and the DIF coset inverse"""
    a = cref.gen_scalars(1 << 24, 7100 + flags, 0)
    assert np.array_equal(ctx.ntt(a, 24, flags), cref.ntt(a, 24, flags))


def test_generic_g1_msm_2p26_pairs_equals_oracle(ctx):
    ctx.trim()   # what the earlier sizes grew (14 GB of NTT vectors and tables at 2^26) goes back first: mi_ctx_trim
    n = 1 << 26
    pts = ctx.gen_g1(n, 4242); sc = ctx.gen_scalars(n, 2424, 0)
    got = ctx.msm_g1_dev(pts.ptr, sc.ptr, n)
    st = ctx.stats()
    hp, hs = pts.download((n, 8)), sc.download((n, 4))
    pts.free(); sc.free()
    assert np.array_equal(hp[:2048], cref.gen_g1(2048, 4242)) and cref.g1_on_curve(hp[-50000:])
    want = _timed(f"oracle G1 MSM, 2^26 pairs (GPU: {st['total_ms']:.0f} ms)", lambda: cref.msm_g1(hp, hs))
    assert np.array_equal(got, want)


def test_g2_msm_2p25_pairs_equals_oracle(ctx):
    ctx.trim()
    n = 1 << 25
    pts = ctx.gen_g2(n, 4343); sc = ctx.gen_scalars(n, 3434, 1)
    got = ctx.msm_g2_dev(pts.ptr, sc.ptr, n)
    hp, hs = pts.download((n, 16)), sc.download((n, 4))
    pts.free(); sc.free()
    assert np.array_equal(hp[:256], cref.gen_g2(256, 4343)) and cref.g2_on_curve(hp[-20000:])
    want = _timed("oracle G2 MSM, 2^25 pairs", lambda: cref.msm_g2(hp, hs))
    assert np.array_equal(got, want)
