"""CPU suite: the device arithmetic headers (gnark-whir_amd/csrc/field.cuh, curve.cuh) compiled
for the host and compared with the oracle.  The product never runs this build; it exists so that
limb-level arithmetic is proven before a kernel is launched on a GPU."""
import ctypes as C
import os
import subprocess
import numpy as np
import pytest
import pyref as P
import cref
from helpers import *

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def emu():
    so = os.path.join(HERE, "emu", "libemu.so")
    src = os.path.join(HERE, "emu", "emu.cpp")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-DMI_CHECK_NOWRAP", "-shared", "-fPIC", "-o", so, src])
    return C.CDLL(so)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def test_field_ops_bit_exact(emu):
    rng = P.SplitMix64(21)
    for field, mod, arr in ((0, P.R_MOD, fr_arr), (1, P.Q_MOD, fp_arr)):
        xs = [rng.fr() % mod for _ in range(300)] + [0, 1, mod - 1, mod - 2, 2, 0, mod - 1]
        ys = [rng.fr() % mod for _ in range(300)] + [0, mod - 1, mod - 1, 1, mod - 2, 5, 1]
        X, Y = arr(xs), arr(ys)
        for op in range(6):
            Z = np.zeros_like(X)
            emu.emu_field_op(field, op, _p(Z), _p(X), _p(Y), C.c_size_t(len(xs)))
            assert np.array_equal(Z, cref.field_op(field, op, X, Y)), (field, op)
        Z = np.zeros_like(X)
        emu.emu_field_op(field, 6, _p(Z), _p(X), _p(Y), C.c_size_t(len(xs)))
        assert np.array_equal(Z, cref.field_op(field, 1, np.zeros_like(X), X))


def adversarial_mont_pairs(mod, seed, n_per=40):
    """Raw Montgomery-form residues (< mod) whose 32-bit limbs hit 0xFFFFFFFF where a column's first product sits:
    x.l[0] and y.l[k], k = 1..6 -- the vectors that broke round 1's carry-less first product (field.cuh:259)
    -- plus all-ones low limbs on both sides."""
    rng = P.SplitMix64(seed)
    xs, ys = [], []
    top = mod >> 224
    def rnd():
        v = rng.fr()
        return (v & ((1 << 224) - 1)) | ((rng.next() % top) << 224)
    for k in range(1, 7):
        for _ in range(n_per):
            x = rnd() | 0xFFFFFFFF
            y = rnd() | (0xFFFFFFFF << (32 * k))
            xs += [x, y]; ys += [y, x]
    for _ in range(n_per):
        x = rnd() | ((1 << 224) - 1); y = rnd() | ((1 << 224) - 1)
        xs.append(x); ys.append(y)
    assert all(v < mod for v in xs + ys)
    return xs, ys


def raw_arr(vals):
    return np.array([cref.int_to_limbs(v) for v in vals], dtype=np.uint64).reshape(-1, 4)


def test_montgomery_product_carry_edge_limbs(emu):
    """operator*, fe_mul2_add, fe_mul_sub, fe_sqr on limbs of 0xFFFFFFFF against Python big integers (and the C oracle):
    every product is x*y/R mod p exactly; the emu build traps if a carry-less first product ever wraps."""
    Rinv = {P.R_MOD: pow(1 << 256, -1, P.R_MOD), P.Q_MOD: pow(1 << 256, -1, P.Q_MOD)}
    for field, mod in ((0, P.R_MOD), (1, P.Q_MOD)):
        xs, ys = adversarial_mont_pairs(mod, 77 + field)
        X, Y = raw_arr(xs), raw_arr(ys)
        ri = Rinv[mod]
        want = {2: [x * y * ri % mod for x, y in zip(xs, ys)],
                7: [2 * x * y * ri % mod for x, y in zip(xs, ys)],
                8: [(x * y - y * y) * ri % mod for x, y in zip(xs, ys)],
                9: [x * x * ri % mod for x in xs]}
        for op, w in want.items():
            Z = np.zeros_like(X)
            emu.emu_field_op(field, op, _p(Z), _p(X), _p(Y), C.c_size_t(len(xs)))
            got = [cref.limbs_to_int(r) for r in Z]
            assert got == w, (field, op, sum(a != b for a, b in zip(got, w)))
        assert np.array_equal(cref.field_op(field, 2, X, Y), raw_arr(want[2]))


def test_limb29_field_roundtrip_and_product(emu):
    """field29.cuh: standard Montgomery form -> nine 29-bit limbs (R' = 2^261) -> back, and the carry-free product, for Fp and Fr;
    random and edge values (0, 1, p - 1, all-ones low limbs)"""
    rng = P.SplitMix64(91)
    for field, mod, arr, op_field in ((1, P.Q_MOD, fp_arr, 1), (0, P.R_MOD, fr_arr, 0)):
        vals = [rng.fr() % mod for _ in range(200)] + [0, 1, mod - 1, mod - 2, (1 << 253) - 1, (1 << 224) - 1, 2, mod - 1]
        X = arr(vals); Y = arr(vals[::-1])
        inp = np.empty((2 * len(vals), 4), np.uint64); inp[0::2] = X; inp[1::2] = Y
        out = np.zeros_like(inp)
        emu.emu_f29_roundtrip(_p(out), _p(inp), C.c_size_t(len(vals)), field)
        assert np.array_equal(out[0::2], X)
        assert np.array_equal(out[1::2], cref.field_op(op_field, 2, X, Y))


def test_limb29_primitives_at_their_documented_bounds(emu):
    """field29.cuh primitives on limb vectors pushed to the edges of their contracts (limbs up to 2^30 / 2^31 - 1 where allowed,
    values up to 8p x 8p), against Python big integers; the emu build traps on any 64-bit column overflow or limb underflow"""
    M29 = (1 << 29) - 1
    rng = np.random.default_rng(17)
    def val(l):
        return sum(int(x) << (29 * i) for i, x in enumerate(l))
    def limbs_of(v):
        return [(v >> (29 * i)) & M29 for i in range(8)] + [v >> 232]
    def call(field, op, a, b=None, c=None, d=None):
        z = [0] * 9
        arrs = [np.array(x if x is not None else z, dtype=np.uint32) for x in (a, b, c, d)]
        out = np.zeros(9, np.uint32)
        assert emu.emu_f29_prim(field, op, _p(out), *[_p(x) for x in arrs]) == 0
        return [int(x) for x in out]
    for field, mod in ((1, P.Q_MOD), (0, P.R_MOD)):
        Rinv = pow(1 << 261, -1, mod)
        for trial in range(60):
            # products: operands up to 8p, limbs up to 2^30 on one side (sum of two weak numbers)
            x = int(rng.integers(0, 1 << 62)) * mod // (1 << 59) % (8 * mod); y = int(rng.integers(0, 1 << 62)) * mod // (1 << 59) % (8 * mod)
            if trial < 8:
                x, y = [(8 * mod - 1, 8 * mod - 1), (0, 5), (mod, mod), (8 * mod - 1, 1), ((1 << 257) - 1, (1 << 257) - 1), (1, 1), (mod - 1, mod + 1), (7 * mod + 12345, 3)][trial]
            lx, ly = limbs_of(x), limbs_of(y)
            if trial % 3 == 1:   # un-normalised: split x = u + v, add limb-wise (limbs up to 2^30)
                u = x // 2; lx = [p_ + q_ for p_, q_ in zip(limbs_of(u), limbs_of(x - u))]
            got = call(field, 0, lx, ly)
            assert val(got) % mod == x * y * Rinv % mod and all(g <= M29 for g in got[:8]) and val(got) < (x * y // (1 << 261)) + mod + 1
            if field == 1:   # the 45-product square: the same limbs as the product with itself
                assert call(1, 8, lx) == call(1, 0, lx, lx)
            # dual product
            u_, v_ = (x * 7 + 3) % (8 * mod), (y * 5 + 1) % (8 * mod)
            got = call(field, 1, limbs_of(x), limbs_of(y), limbs_of(u_), limbs_of(v_))
            assert val(got) % mod == (x * y + u_ * v_) * Rinv % mod
        if field == 1:
            for K, op in ((8, 2), (4, 3), (2, 4)):
                for trial in range(40):
                    xv = int(rng.integers(0, 1 << 62)) * mod // (1 << 60) % (8 * mod)
                    yv = int(rng.integers(0, 1 << 62)) * mod // (1 << 62) % (K * mod - (1 << 233))
                    ly = limbs_of(yv)
                    if trial % 2:   # weakly normalised subtrahend: limbs up to 2^29 + 7
                        for i in range(8):
                            if ly[i + 1] > 0 and ly[i] + (1 << 29) <= M29 + 8: ly[i + 1] -= 1; ly[i] += 1 << 29
                    got = call(1, op, limbs_of(xv), ly)
                    assert val(got) == xv + K * mod - val(ly) and all(g < (1 << 31) for g in got)
                    wn = call(1, 5, got)
                    assert val(wn) == val(got) and all(g <= M29 + 8 for g in wn[:8])
            for K, op in ((4, 6), (2, 7)):
                for v in [0, mod, K * mod - 1, K * mod, K * mod + (1 << 233), 2 * K * mod - 1, (K + 1) * mod, int(1.5 * K * mod)]:
                    got = call(1, op, limbs_of(v))
                    r = val(call(1, 5, got))
                    assert r % mod == v % mod and r <= max(v - K * mod, K * mod + (1 << 233)) and r >= 0 and r < K * mod + (1 << 234)


def test_limb29_mixed_addition_chain_matches_oracle(emu):
    """curve29.cuh: chains of mixed additions in 29-bit limbs (lazy bounds hand-tracked; the host build traps on any limb
    underflow or 64-bit column overflow) against the oracle's group law -- random chains with sign flips, the same point twice
    (doubling through the standard-arithmetic slow path), a point and its opposite (cancellation to infinity and restart),
    infinity entries, and a long chain"""
    g = cref.gen_g1(400, 11)
    def run(pts, neg):
        out = np.zeros(8, np.uint64)
        pts = np.ascontiguousarray(pts); neg = np.ascontiguousarray(neg, dtype=np.uint8)
        emu.emu_g1_madd29_chain(_p(out), _p(pts), neg.ctypes.data_as(C.c_char_p), C.c_size_t(pts.shape[0]))
        return out
    def want(pts, neg):
        acc = None
        for p_, n_ in zip(g1_pts(pts), neg):
            q = P.g1_neg(p_) if (n_ and p_ is not None) else p_
            acc = q if acc is None else P.g1_add(acc, q)
        return g1_arr([acc])[0]
    rng = np.random.default_rng(3)
    for n in (1, 2, 3, 16, 64, 400):
        neg = rng.integers(0, 2, n)
        assert np.array_equal(run(g[:n], neg), want(g[:n], neg)), n
    dbl = np.stack([g[0], g[0], g[1], g[1], g[1]]); z = np.zeros(5, np.uint8)
    assert np.array_equal(run(dbl, z), want(dbl, z))
    canc = np.stack([g[2], g[2], g[3], np.zeros(8, np.uint64), g[4]]); ng = np.array([0, 1, 0, 0, 1], np.uint8)
    assert np.array_equal(run(canc, ng), want(canc, ng))
    allc = np.stack([g[5], g[5]]); assert not run(allc, np.array([0, 1], np.uint8)).any()


def test_limb29_partial_sums_in_rprime_form_match_oracle(emu):
    """curve29.cuh g1x29_store_rp / g1x29_load_rp / g1x29_add: items of mixed additions whose sums stay packed in the R' form and are
    added level by level (what k_msm_accum_affine29 + k_msm_accum_xyzz29 do), against the oracle's group law -- including equal
    partial sums (doubling), opposite ones (cancellation), infinite ones, and items of one point"""
    g = cref.gen_g1(300, 12)
    def run(pts, neg, item_len):
        out = np.zeros(8, np.uint64)
        pts = np.ascontiguousarray(pts); neg = np.ascontiguousarray(neg, dtype=np.uint8)
        emu.emu_g1_rp_levels(_p(out), _p(pts), neg.ctypes.data_as(C.c_char_p), C.c_size_t(pts.shape[0]), C.c_size_t(item_len))
        return out
    def want(pts, neg):
        acc = None
        for p_, n_ in zip(g1_pts(pts), neg):
            q = P.g1_neg(p_) if (n_ and p_ is not None) else p_
            acc = q if acc is None else P.g1_add(acc, q)
        return g1_arr([acc])[0]
    rng = np.random.default_rng(4)
    for n, L in ((1, 1), (2, 1), (7, 2), (64, 4), (300, 16), (300, 1), (100, 7)):
        neg = rng.integers(0, 2, n)
        assert np.array_equal(run(g[:n], neg, L), want(g[:n], neg)), (n, L)
    z4 = np.zeros(4, np.uint8)
    same = np.stack([g[0], g[1], g[0], g[1]])                      # two equal partial sums: doubling through the slow path
    assert np.array_equal(run(same, z4, 2), want(same, z4))
    opp = np.stack([g[0], g[1], g[0], g[1]]); ng = np.array([0, 0, 1, 1], np.uint8)   # opposite partial sums: infinity
    assert not run(opp, ng, 2).any()
    infs = np.stack([g[2], g[2], g[3], g[4]]); ni = np.array([0, 1, 0, 0], np.uint8)  # first item sums to infinity
    assert np.array_equal(run(infs, ni, 2), want(infs, ni))
    zer = np.stack([np.zeros(8, np.uint64), np.zeros(8, np.uint64), g[5], g[6]])      # an item of infinity entries only
    assert np.array_equal(run(zer, z4, 2), want(zer, z4))


def test_limb29_g2_partial_sums_in_rprime_form_match_oracle(emu):
    """curve29_g2.cuh f2_29_pack / g2x29_add (overflow traps on): items of G2 mixed additions whose sums stay packed in the R' form and
    are added level by level (k_msm_accum_affine_g2_29 + k_msm_accum_xyzz_g2_29), against the oracle's group law -- including equal
    partial sums (doubling), opposite ones (cancellation), infinite ones, items of one point"""
    g = cref.gen_g2(120, 21)
    def run(pts, neg, item_len):
        out = np.zeros(16, np.uint64)
        pts = np.ascontiguousarray(pts); neg = np.ascontiguousarray(neg, dtype=np.uint8)
        emu.emu_g2_rp_levels(_p(out), _p(pts), neg.ctypes.data_as(C.c_char_p), C.c_size_t(pts.shape[0]), C.c_size_t(item_len))
        return out
    def want(pts, neg):
        acc = None
        for p_, n_ in zip(g2_pts(pts), neg):
            q = P.g2_neg(p_) if (n_ and p_ is not None) else p_
            acc = q if acc is None else P.g2_add(acc, q)
        return g2_arr([acc])[0]
    rng = np.random.default_rng(14)
    for n, L in ((1, 1), (2, 1), (7, 2), (64, 4), (120, 16), (60, 1), (100, 7)):
        neg = rng.integers(0, 2, n)
        assert np.array_equal(run(g[:n], neg, L), want(g[:n], neg)), (n, L)
    z4 = np.zeros(4, np.uint8)
    same = np.stack([g[0], g[1], g[0], g[1]])                      # two equal partial sums: doubling through the slow path
    assert np.array_equal(run(same, z4, 2), want(same, z4))
    opp = np.stack([g[0], g[1], g[0], g[1]]); ng = np.array([0, 0, 1, 1], np.uint8)   # opposite partial sums: infinity
    assert not run(opp, ng, 2).any()
    infs = np.stack([g[2], g[2], g[3], g[4]]); ni = np.array([0, 1, 0, 0], np.uint8)  # first item sums to infinity
    assert np.array_equal(run(infs, ni, 2), want(infs, ni))
    zer = np.stack([np.zeros(16, np.uint64), np.zeros(16, np.uint64), g[5], g[6]])    # an item of infinity entries only
    assert np.array_equal(run(zer, z4, 2), want(zer, z4))


def test_limb29_g2_mixed_addition_chain_matches_oracle(emu):
    """curve29_g2.cuh on the host (overflow traps on): Fp2 over 29-bit limbs, dual products, the conditional -4p / -2p of X3; chains
    with sign flips, doubling, cancellation, infinity entries"""
    g = cref.gen_g2(80, 12)
    def run(pts, neg):
        out = np.zeros(16, np.uint64)
        pts = np.ascontiguousarray(pts); neg = np.ascontiguousarray(neg, dtype=np.uint8)
        emu.emu_g2_madd29_chain(_p(out), _p(pts), neg.ctypes.data_as(C.c_char_p), C.c_size_t(pts.shape[0]))
        return out
    def want(pts, neg):
        acc = None
        for p_, n_ in zip(g2_pts(pts), neg):
            q = P.g2_neg(p_) if (n_ and p_ is not None) else p_
            acc = q if acc is None else P.g2_add(acc, q)
        return g2_arr([acc])[0]
    rng = np.random.default_rng(4)
    for n in (1, 2, 3, 17, 80):
        neg = rng.integers(0, 2, n)
        assert np.array_equal(run(g[:n], neg), want(g[:n], neg)), n
    dbl = np.stack([g[0], g[0], g[1], g[1], g[1]]); z = np.zeros(5, np.uint8)
    assert np.array_equal(run(dbl, z), want(dbl, z))
    canc = np.stack([g[2], g[2], g[3], np.zeros(16, np.uint64), g[4]]); ng = np.array([0, 1, 0, 0, 1], np.uint8)
    assert np.array_equal(run(canc, ng), want(canc, ng))
    assert not run(np.stack([g[5], g[5]]), np.array([0, 1], np.uint8)).any()


def test_curve_ops_bit_exact(emu):
    g1 = cref.gen_g1(64, 5)
    a = np.concatenate([g1[:32], g1[:4], g1[4:8], np.zeros((2, 8), np.uint64), g1[8:9]])
    neg = g1_arr([P.g1_neg(p) for p in g1_pts(g1[4:8])])
    b = np.concatenate([g1[32:], g1[:4], neg, g1[9:10], np.zeros((1, 8), np.uint64), np.zeros((1, 8), np.uint64)])
    want = cref.g1_add(a, b)
    for mode in (0, 1):
        out = np.zeros_like(a)
        emu.emu_g1_add(_p(out), _p(a), _p(b), C.c_size_t(a.shape[0]), mode)
        assert np.array_equal(out, want), mode
    out = np.zeros_like(a)
    emu.emu_g1_add(_p(out), _p(a), _p(b), C.c_size_t(a.shape[0]), 2)
    nb = g1_arr([P.g1_neg(p) for p in g1_pts(b)])
    assert np.array_equal(out, cref.g1_add(a, nb))
    g2 = cref.gen_g2(24, 6)
    a2 = np.concatenate([g2[:8], g2[:2], np.zeros((1, 16), np.uint64)])
    b2 = np.concatenate([g2[8:16], g2[:2], g2[3:4]])
    want2 = cref.g2_add(a2, b2)
    for mode in (0, 1):
        out = np.zeros_like(a2)
        emu.emu_g2_add(_p(out), _p(a2), _p(b2), C.c_size_t(a2.shape[0]), mode)
        assert np.array_equal(out, want2), mode
    o = np.zeros(8, np.uint64)
    emu.emu_g1_mul_u32(_p(o), _p(g1[0].copy()), 0xDEADBEEF)
    assert g1_pts(o) == [P.g1_mul(g1_pts(g1[:1])[0], 0xDEADBEEF)]
    b2c = np.zeros(8, np.uint64)
    emu.emu_g2_b(_p(b2c))
    assert tuple(fp_vals(b2c.reshape(2, 4))) == P.G2_B


@pytest.mark.parametrize("log_n,log_e,mc,ms", [(1, 11, 11, 8), (4, 11, 11, 8), (6, 3, 3, 2), (9, 5, 4, 3), (10, 6, 3, 3), (12, 11, 11, 8), (13, 7, 5, 4)])
def test_ntt_pass_decomposition_matches_oracle(emu, log_n, log_e, mc, ms):
    """ntt_tile.cuh (the kernel's index arithmetic) run on the host with small tiles / radices so
    that 1-, 2-, 3- and 4-pass plans are all exercised, against the C oracle, all 8 modes."""
    n = 1 << log_n
    a = cref.gen_scalars(n, 100 + log_n, 0)
    for flags in range(8):
        got = a.copy()
        npass = emu.emu_ntt(_p(got), log_n, flags, log_e, mc, ms, 64, n)
        assert npass >= 1
        assert np.array_equal(got, cref.ntt(a, log_n, flags)), (flags, npass)


def test_ntt_fused_zero_padding(emu):
    log_n, n_valid = 8, 150
    a = cref.gen_scalars(1 << log_n, 3, 0)
    padded = a.copy(); padded[n_valid:] = 0
    got = a.copy()
    emu.emu_ntt(_p(got), log_n, 1, 5, 4, 2, 64, n_valid)
    assert np.array_equal(got, cref.ntt(padded, log_n, 1))


@pytest.mark.parametrize("log_n,log_e,mc,ms", [(4, 11, 11, 8), (9, 5, 4, 3), (10, 6, 3, 3)])
def test_ntt_fused_pointwise_edges(emu, log_n, log_e, mc, ms):
    """NttPass::load_mul / store_sub (computeH's last transform): FFTInverse on the coset of x * m, minus s, in one transform"""
    n = 1 << log_n
    x, m, s = (cref.gen_scalars(n, 7 + k, 0) for k in range(3))
    got = x.copy()
    emu.emu_ntt_fused(_p(got), log_n, 3, log_e, mc, ms, 64, n, _p(m), _p(s))
    assert np.array_equal(got, cref.field_op(0, 1, cref.ntt(cref.field_op(0, 2, x, m), log_n, 3), s))


@pytest.mark.parametrize("log_n,n_constraints", [(3, 8), (8, 200), (12, 4000)])
def test_compute_h_needs_six_transforms(log_n, n_constraints):
    """The identity the device's computeH rests on, with the oracle's own transforms: the coset FFT of c is undone by the last
    (linear) transform, so  h = den * FFTInverse_coset(ca * cb) - den * FFTInverse(c)  for ANY a, b, c (c is NOT a * b here)."""
    n = 1 << log_n
    a, b, c = (cref.gen_scalars(n_constraints, 40 + k, k % 2) for k in range(3))
    pad = lambda v: np.concatenate([v, np.zeros((n - len(v), 4), np.uint64)])
    want = cref.compute_h(log_n, a, b, c)
    ia, ib, ic = (cref.ntt(pad(v), log_n, 1) for v in (a, b, c))
    ca, cb = cref.ntt(ia, log_n, 2 | 4), cref.ntt(ib, log_n, 2 | 4)
    r = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
    den = pow(pow(5, n, r) - 1, -1, r)
    den_m = np.tile(np.array([[(den * (1 << 256) % r) >> (64 * k) & (2 ** 64 - 1) for k in range(4)]], np.uint64), (n, 1))
    x = cref.ntt(cref.field_op(0, 2, ca, cb), log_n, 1 | 2)
    got = cref.field_op(0, 1, cref.field_op(0, 2, x, den_m), cref.field_op(0, 2, ic, den_m))
    assert np.array_equal(got, want)


@pytest.mark.parametrize("n,dist,c,G,L,seg", [(1, 0, 4, 1, 4, 2), (2, 1, 3, 2, 2, 3), (37, 0, 5, 3, 4, 4), (300, 1, 4, 4, 3, 2),
                                              (300, 0, 8, 2, 32, 16), (1000, 1, 16, 3, 8, 64), (257, 1, 7, 5, 2, 8)])
def test_msm_pipeline_matches_oracle(emu, n, dist, c, G, L, seg):
    """msm_core.cuh (digits, counting sort, item/level accumulation, bucket reduce, window combine)
    run on the host with small windows / items so multi-level paths are exercised."""
    pts = cref.gen_g1(n, 500 + n); sc = cref.gen_scalars(n, 600 + n, dist)
    if n > 30:
        pts[3] = 0; sc[1] = fr_arr([P.R_MOD - 1])[0]; sc[2] = 0; pts[6] = pts[5]; sc[6] = sc[5]
        pts[8] = g1_arr([P.g1_neg(g1_pts(pts[7:8])[0])])[0]; sc[8] = sc[7]
    out = np.zeros(8, np.uint64)
    levels = emu.emu_msm_g1(_p(out), _p(pts), _p(sc), n, 1, c, G, L, seg, 7)
    want = cref.msm_g1(pts, sc)
    assert np.array_equal(out, want[:8]) or (want[8:].any() == 0 and not out.any()), levels
    if n == 300 and dist == 1:
        assert levels >= 3   # the heavy {0,1} bucket needs several levels at L = 3
    canon = cref.field_op(0, 5, sc)
    out2 = np.zeros(8, np.uint64)
    emu.emu_msm_g1(_p(out2), _p(pts), _p(canon), n, 0, c, G, L, seg, 7)
    assert np.array_equal(out2, out)


def test_msm_pipeline_g2_matches_oracle(emu):
    n = 60
    pts = cref.gen_g2(n, 9); sc = cref.gen_scalars(n, 10, 1)
    out = np.zeros(16, np.uint64)
    emu.emu_msm_g2(_p(out), _p(pts), _p(sc), n, 1, 5, 2, 3, 4, 5)
    assert np.array_equal(out, cref.msm_g2(pts, sc)[:16])


@pytest.mark.parametrize("n,dist,c,G,chunk,L,gbits", [(1, 0, 17, 1, 8, 4, 15), (60, 1, 17, 3, 16, 3, 11), (200, 0, 18, 2, 50, 8, 9),
                                                     (150, 1, 19, 4, 7, 4, 12), (90, 1, 17, 2, 16, 4, 16 - 1)])
def test_fixed_base_msm_pipeline_matches_oracle(emu, n, dist, c, G, chunk, L, gbits):
    """msm2_core.cuh: window copies 2^(c*w)*P, one bucket set for all windows, two-pass sort (partition by the high
    bucket bits, chunked LDS counting sort by the low gbits), then the shared item / bucket-reduce bodies"""
    pts = cref.gen_g1(n, 900 + n); sc = cref.gen_scalars(n, 901 + n, dist)
    if n > 30:
        pts[3] = 0; sc[1] = fr_arr([P.R_MOD - 1])[0]; sc[2] = 0; pts[6] = pts[5]; sc[6] = sc[5]
    out = np.zeros(8, np.uint64)
    nchunks = emu.emu_msm2_g1(_p(out), _p(pts), _p(sc), n, 1, c, G, chunk, L, 64, 5, gbits, 0)
    want = cref.msm_g1(pts, sc)
    assert nchunks >= 1 and (np.array_equal(out, want[:8]) or (not want[8:].any() and not out.any()))


@pytest.mark.parametrize("n,dist,c,G,chunk,L,gbits", [(1, 0, 4, 1, 8, 4, 3), (70, 1, 5, 3, 16, 3, 4), (300, 0, 8, 2, 50, 8, 6), (257, 1, 6, 4, 7, 4, 5)])
def test_generic_msm_through_two_pass_sort_matches_oracle(emu, n, dist, c, G, chunk, L, gbits):
    """msm2_core.cuh with the window folded into the key (wkeys): what large generic MSMs (c = 16) use instead of the one-pass
    scatter -- key = (w << (c-1)) | bucket, value = point index | sign, per-window bucket sets, window combine"""
    pts = cref.gen_g1(n, 950 + n); sc = cref.gen_scalars(n, 951 + n, dist)
    if n > 30:
        pts[3] = 0; sc[1] = fr_arr([P.R_MOD - 1])[0]; sc[2] = 0; pts[6] = pts[5]; sc[6] = sc[5]
    out = np.zeros(8, np.uint64)
    nchunks = emu.emu_msm2_g1(_p(out), _p(pts), _p(sc), n, 1, c, G, chunk, L, 4, 5, gbits, 1)
    want = cref.msm_g1(pts, sc)
    assert nchunks >= 1 and (np.array_equal(out, want[:8]) or (not want[8:].any() and not out.any()))
