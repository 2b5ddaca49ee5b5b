"""Definitional big-integer oracle for the Groth16 prove path on BN254.

TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import or execute it.

PARITY UNPINNED.  The arithmetic the reference runs at /root/reference/mt.go:496
(groth16.Prove) lives in un-vendored third-party modules that are absent from the
container (github.com/consensys/gnark v0.11.0, go.mod:6;
github.com/consensys/gnark-crypto v0.14.1-0.20241217131346-b998989abdbe, go.mod:7), the
reference holds no tests / golden vectors / fixtures, and there is
no Go toolchain here.  This file therefore restates the *published algorithm* of those
pinned versions from the mathematics:

  * Fr / Fp are the BN254 scalar / base fields; the modulus r is the one cited at
    /root/reference/typeConverters/typeConverters.go:28, the 4 x u64 little-endian limb
    order is the one documented at typeConverters.go:30-39.
  * NTT conventions follow gnark-crypto ecc/bn254/fr/fft (Domain, FFT, FFTInverse,
    DIF = natural in / bit-reversed out, DIT = bit-reversed in / natural out,
    OnCoset with shift g = 5) by behaviour.
  * compute_h / prove follow gnark backend/groth16/bn254/prove.go by behaviour
    (SURVEY.md section 3.3), the API order is that of mt.go:447-497.
  * point encoding follows gnark-crypto ecc/bn254/marshal.go by behaviour
    (SURVEY.md section 8a row a12).

What pins it instead: every function here is checked against an independent
definition in tests/test_oracle.py (O(n^2) DFT, double-and-add MSM, the Groth16
verification equation evaluated in the exponent with the known toy trapdoor).
"""
from __future__ import annotations

# --------------------------------------------------------------------------- constants
R_MOD = 21888242871839275222246405745257275088548364400416034343698204186575808495617  # Fr
Q_MOD = 21888242871839275222246405745257275088696311157297823662689037894645226208583  # Fp
MONT_R = 1 << 256
FR_ROOT_2_28 = 19103219067921713944291392827692070036145651957329286315305642004821462161904
FR_TWO_ADICITY = 28
FR_COSET_GEN = 5
G1_B = 3
G1_GEN = (1, 2)
# Fp2 = Fp[u]/(u^2+1); elements are (a0, a1) = a0 + a1*u
G2_GEN = (
    (10857046999023057135944570762232829481370756359578518086990519993285655852781,
     11559732032986387107991004021392285783925812861821192530917403151452391805634),
    (8495653923123431417604973247489272438418190587263600148770280649306958101930,
     4082367875863433681332203403145435568316851327593401208105741076214120093531),
)


def fr_inv(a: int) -> int:
    return pow(a, R_MOD - 2, R_MOD)


def fp_inv(a: int) -> int:
    return pow(a, Q_MOD - 2, Q_MOD)


# --------------------------------------------------------------------------- limb packing
def to_limbs(x: int) -> bytes:
    """32-byte little-endian = 4 x u64 LE limbs (typeConverters.go:30-39 order)."""
    return int(x).to_bytes(32, "little")


def from_limbs(b: bytes) -> int:
    return int.from_bytes(b, "little")


def fr_to_mont(x: int) -> int:
    return (x * MONT_R) % R_MOD


def fr_from_mont(x: int) -> int:
    return (x * pow(MONT_R, -1, R_MOD)) % R_MOD


def fp_to_mont(x: int) -> int:
    return (x * MONT_R) % Q_MOD


def fp_from_mont(x: int) -> int:
    return (x * pow(MONT_R, -1, Q_MOD)) % Q_MOD


# --------------------------------------------------------------------------- Fp2
def fp2_add(a, b):
    return ((a[0] + b[0]) % Q_MOD, (a[1] + b[1]) % Q_MOD)


def fp2_sub(a, b):
    return ((a[0] - b[0]) % Q_MOD, (a[1] - b[1]) % Q_MOD)


def fp2_neg(a):
    return ((-a[0]) % Q_MOD, (-a[1]) % Q_MOD)


def fp2_mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % Q_MOD, (a[0] * b[1] + a[1] * b[0]) % Q_MOD)


def fp2_sqr(a):
    return fp2_mul(a, a)


def fp2_inv(a):
    n = fp_inv((a[0] * a[0] + a[1] * a[1]) % Q_MOD)
    return ((a[0] * n) % Q_MOD, (-a[1] * n) % Q_MOD)


def fp2_scalar(a, k: int):
    return ((a[0] * k) % Q_MOD, (a[1] * k) % Q_MOD)


FP2_ZERO = (0, 0)
FP2_ONE = (1, 0)
# twist: y^2 = x^3 + 3/(9+u)
G2_B = fp2_mul((3, 0), fp2_inv((9, 1)))


# --------------------------------------------------------------------------- generic curve
class _Field:
    """Tiny field vtable so that G1 (ints mod q) and G2 (Fp2 tuples) share the group law."""

    def __init__(self, add, sub, mul, inv, neg, zero, one, is_zero):
        self.add, self.sub, self.mul, self.inv, self.neg = add, sub, mul, inv, neg
        self.zero, self.one, self.is_zero = zero, one, is_zero


F1 = _Field(lambda a, b: (a + b) % Q_MOD, lambda a, b: (a - b) % Q_MOD, lambda a, b: (a * b) % Q_MOD,
            fp_inv, lambda a: (-a) % Q_MOD, 0, 1, lambda a: a % Q_MOD == 0)
F2 = _Field(fp2_add, fp2_sub, fp2_mul, fp2_inv, fp2_neg, FP2_ZERO, FP2_ONE, lambda a: a == FP2_ZERO)

# Affine points are (x, y) or None for infinity (gnark encodes infinity as (0,0)).


def ec_add(F: _Field, P, Q):
    """Affine chord-and-tangent addition on y^2 = x^3 + b (a = 0)."""
    if P is None:
        return Q
    if Q is None:
        return P
    x1, y1 = P
    x2, y2 = Q
    if x1 == x2:
        if F.is_zero(F.add(y1, y2)):
            return None
        # doubling
        num = F.mul(F.add(F.add(F.mul(x1, x1), F.mul(x1, x1)), F.mul(x1, x1)), F.one)
        lam = F.mul(num, F.inv(F.add(y1, y1)))
    else:
        lam = F.mul(F.sub(y2, y1), F.inv(F.sub(x2, x1)))
    x3 = F.sub(F.sub(F.mul(lam, lam), x1), x2)
    y3 = F.sub(F.mul(lam, F.sub(x1, x3)), y1)
    return (x3, y3)


def ec_neg(F: _Field, P):
    return None if P is None else (P[0], F.neg(P[1]))


# Jacobian internals for speed (x = X/Z^2, y = Y/Z^3), a = 0.
def _jac_dbl(F, P):
    X, Y, Z = P
    if F.is_zero(Z):
        return P
    A = F.mul(X, X)
    B = F.mul(Y, Y)
    C = F.mul(B, B)
    t = F.add(X, B)
    D = F.sub(F.sub(F.mul(t, t), A), C)
    D = F.add(D, D)
    E = F.add(F.add(A, A), A)
    Fq = F.mul(E, E)
    X3 = F.sub(Fq, F.add(D, D))
    C8 = F.add(C, C)
    C8 = F.add(C8, C8)
    C8 = F.add(C8, C8)
    Y3 = F.sub(F.mul(E, F.sub(D, X3)), C8)
    Z3 = F.mul(F.add(Y, Y), Z)
    return (X3, Y3, Z3)


def _jac_add(F, P, Q):
    X1, Y1, Z1 = P
    X2, Y2, Z2 = Q
    if F.is_zero(Z1):
        return Q
    if F.is_zero(Z2):
        return P
    Z1Z1 = F.mul(Z1, Z1)
    Z2Z2 = F.mul(Z2, Z2)
    U1 = F.mul(X1, Z2Z2)
    U2 = F.mul(X2, Z1Z1)
    S1 = F.mul(F.mul(Y1, Z2), Z2Z2)
    S2 = F.mul(F.mul(Y2, Z1), Z1Z1)
    if U1 == U2:
        if S1 == S2:
            return _jac_dbl(F, P)
        return (F.one, F.one, F.zero)
    H = F.sub(U2, U1)
    Rr = F.sub(S2, S1)
    HH = F.mul(H, H)
    HHH = F.mul(H, HH)
    V = F.mul(U1, HH)
    X3 = F.sub(F.sub(F.mul(Rr, Rr), HHH), F.add(V, V))
    Y3 = F.sub(F.mul(Rr, F.sub(V, X3)), F.mul(S1, HHH))
    Z3 = F.mul(F.mul(Z1, Z2), H)
    return (X3, Y3, Z3)


def _to_jac(F, P):
    return (F.one, F.one, F.zero) if P is None else (P[0], P[1], F.one)


def _from_jac(F, P):
    X, Y, Z = P
    if F.is_zero(Z):
        return None
    zi = F.inv(Z)
    zi2 = F.mul(zi, zi)
    return (F.mul(X, zi2), F.mul(Y, F.mul(zi2, zi)))


def ec_mul(F: _Field, P, k: int):
    """k*P by left-to-right double-and-add (k taken as a plain non-negative integer)."""
    if P is None or k == 0:
        return None
    assert k >= 0
    acc = (F.one, F.one, F.zero)
    base = _to_jac(F, P)
    for bit in bin(k)[2:]:
        acc = _jac_dbl(F, acc)
        if bit == "1":
            acc = _jac_add(F, acc, base)
    return _from_jac(F, acc)


def ec_sum(F: _Field, pts):
    acc = (F.one, F.one, F.zero)
    for P in pts:
        acc = _jac_add(F, acc, _to_jac(F, P))
    return _from_jac(F, acc)


def g1_is_on_curve(P) -> bool:
    if P is None:
        return True
    x, y = P
    return (y * y - x * x * x - G1_B) % Q_MOD == 0


def g2_is_on_curve(P) -> bool:
    if P is None:
        return True
    x, y = P
    return fp2_sub(fp2_sqr(y), fp2_add(fp2_mul(fp2_sqr(x), x), G2_B)) == FP2_ZERO


def g1_add(P, Q): return ec_add(F1, P, Q)
def g2_add(P, Q): return ec_add(F2, P, Q)
def g1_mul(P, k): return ec_mul(F1, P, k)
def g2_mul(P, k): return ec_mul(F2, P, k)
def g1_neg(P): return ec_neg(F1, P)
def g2_neg(P): return ec_neg(F2, P)


def msm_naive(F: _Field, points, scalars):
    """Definition of MultiExp: sum_i s_i * P_i, s_i canonical integers in [0, r)."""
    acc = (F.one, F.one, F.zero)
    for P, s in zip(points, scalars):
        if P is None or s % R_MOD == 0:
            continue
        acc = _jac_add(F, acc, _to_jac(F, ec_mul(F, P, s % R_MOD)))
    return _from_jac(F, acc)


def msm_pippenger(F: _Field, points, scalars, c: int = 8):
    """Bucket method, unsigned c-bit digits; used only to make mid-size fixtures affordable.
    Checked against msm_naive in tests/test_oracle.py."""
    nwin = (254 + c - 1) // c
    total = (F.one, F.one, F.zero)
    jpts = [_to_jac(F, P) for P in points]
    sc = [s % R_MOD for s in scalars]
    for w in reversed(range(nwin)):
        for _ in range(c):
            total = _jac_dbl(F, total)
        buckets = [None] * (1 << c)
        for P, s in zip(jpts, sc):
            d = (s >> (w * c)) & ((1 << c) - 1)
            if d:
                buckets[d] = P if buckets[d] is None else _jac_add(F, buckets[d], P)
        run = (F.one, F.one, F.zero)
        acc = (F.one, F.one, F.zero)
        for d in range((1 << c) - 1, 0, -1):
            if buckets[d] is not None:
                run = _jac_add(F, run, buckets[d])
            acc = _jac_add(F, acc, run)
        total = _jac_add(F, total, acc)
    return _from_jac(F, total)


# --------------------------------------------------------------------------- NTT (gnark fft.Domain by behaviour)
def bitrev(i: int, logn: int) -> int:
    r = 0
    for _ in range(logn):
        r = (r << 1) | (i & 1)
        i >>= 1
    return r


def bit_reverse_perm(a):
    n = len(a)
    logn = n.bit_length() - 1
    return [a[bitrev(i, logn)] for i in range(n)]


class Domain:
    """fft.NewDomain(n): Generator = root^(2^(28-log n)), FrMultiplicativeGen = 5."""

    def __init__(self, n: int):
        logn = max(n - 1, 0).bit_length()
        self.log_n = logn
        self.n = 1 << logn
        assert logn <= FR_TWO_ADICITY
        self.gen = pow(FR_ROOT_2_28, 1 << (FR_TWO_ADICITY - logn), R_MOD)
        self.gen_inv = fr_inv(self.gen)
        self.card_inv = fr_inv(self.n)
        self.coset_gen = FR_COSET_GEN
        self.coset_gen_inv = fr_inv(FR_COSET_GEN)


def dft_definition(a, w):
    """O(n^2) definition: out[k] = sum_i a[i] w^(ik).  Natural in, natural out."""
    n = len(a)
    return [sum(a[i] * pow(w, (i * k) % n, R_MOD) for i in range(n)) % R_MOD for k in range(n)]


def _dif(a, w):
    """In-place radix-2 decimation-in-frequency: natural in, bit-reversed out."""
    n = len(a)
    a = list(a)
    m = n
    wm = w
    while m >= 2:
        half = m // 2
        for start in range(0, n, m):
            t = 1
            for j in range(half):
                x, y = a[start + j], a[start + j + half]
                a[start + j] = (x + y) % R_MOD
                a[start + j + half] = ((x - y) * t) % R_MOD
                t = (t * wm) % R_MOD
        wm = (wm * wm) % R_MOD
        m = half
    return a


def _dit(a, w):
    """In-place radix-2 decimation-in-time: bit-reversed in, natural out."""
    n = len(a)
    a = list(a)
    m = 2
    while m <= n:
        half = m // 2
        wm = pow(w, n // m, R_MOD)
        for start in range(0, n, m):
            t = 1
            for j in range(half):
                x, y = a[start + j], (a[start + j + half] * t) % R_MOD
                a[start + j] = (x + y) % R_MOD
                a[start + j + half] = (x - y) % R_MOD
                t = (t * wm) % R_MOD
        m *= 2
    return a


DIF, DIT = 0, 1


def fft(dom: Domain, a, decimation: int, coset: bool = False):
    """Domain.FFT: forward transform; with coset, evaluates on g*<w>."""
    n = dom.n
    assert len(a) == n
    a = [x % R_MOD for x in a]
    if coset:
        if decimation == DIT:  # input is bit-reversed: scale slot i by g^bitrev(i)
            a = [(a[i] * pow(dom.coset_gen, bitrev(i, dom.log_n), R_MOD)) % R_MOD for i in range(n)]
        else:
            a = [(a[i] * pow(dom.coset_gen, i, R_MOD)) % R_MOD for i in range(n)]
    return _dif(a, dom.gen) if decimation == DIF else _dit(a, dom.gen)


def fft_inverse(dom: Domain, a, decimation: int, coset: bool = False):
    """Domain.FFTInverse: inverse transform incl. 1/n; with coset also g^-i."""
    n = dom.n
    assert len(a) == n
    a = [x % R_MOD for x in a]
    a = _dif(a, dom.gen_inv) if decimation == DIF else _dit(a, dom.gen_inv)
    if not coset:
        return [(x * dom.card_inv) % R_MOD for x in a]
    if decimation == DIT:  # natural output
        return [(a[i] * pow(dom.coset_gen_inv, i, R_MOD) * dom.card_inv) % R_MOD for i in range(n)]
    return [(a[i] * pow(dom.coset_gen_inv, bitrev(i, dom.log_n), R_MOD) * dom.card_inv) % R_MOD
            for i in range(n)]


def compute_h(a, b, c, dom: Domain):
    """gnark computeH: h = (a*b - c)/Z_H in coefficient form, returned BIT-REVERSED
    (length n; the caller uses h[:n-1])."""
    n = dom.n
    pad = lambda v: list(v) + [0] * (n - len(v))
    a, b, c = pad(a), pad(b), pad(c)
    a = fft_inverse(dom, a, DIF)
    b = fft_inverse(dom, b, DIF)
    c = fft_inverse(dom, c, DIF)
    a = fft(dom, a, DIT, coset=True)
    b = fft(dom, b, DIT, coset=True)
    c = fft(dom, c, DIT, coset=True)
    den = fr_inv((pow(dom.coset_gen, n, R_MOD) - 1) % R_MOD)
    h = [((a[i] * b[i] - c[i]) * den) % R_MOD for i in range(n)]
    return fft_inverse(dom, h, DIF, coset=True)


# --------------------------------------------------------------------------- toy R1CS + Groth16 with known trapdoor
class SplitMix64:
    """Seeded PRNG shared by the oracle, the C restatement and the HIP bench generators."""

    def __init__(self, seed: int):
        self.s = seed & 0xFFFFFFFFFFFFFFFF

    def next(self) -> int:
        self.s = (self.s + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return z ^ (z >> 31)

    def fr(self) -> int:
        v = 0
        for k in range(4):
            v |= self.next() << (64 * k)
        return v % R_MOD

    def below(self, n: int) -> int:
        return self.next() % n


class ToyR1CS:
    """nb_constraints rows, wires = [1, public..., private...]; each row is
    (A_row, B_row, C_row) as sparse dicts wire -> coeff.  Built so a witness exists."""

    def __init__(self, nb_constraints: int, nb_public: int, seed: int, small_frac: float = 0.5):
        rng = SplitMix64(seed)
        self.nb_public = nb_public  # includes the constant wire 1 at index 0
        wires = [1]
        for _ in range(nb_public - 1):
            wires.append(rng.below(256))  # public inputs are bytes (mtUtilities.go:92)
        # a few free private inputs
        n_free = 4
        for _ in range(n_free):
            wires.append(rng.fr())
        rows = []
        for _ in range(nb_constraints):
            kind = rng.below(100)
            def lin():
                d = {}
                for _ in range(1 + rng.below(3)):
                    d[rng.below(len(wires))] = rng.fr() if rng.below(2) else 1 + rng.below(7)
                return d
            if kind < int(small_frac * 100):
                # booleanity-like row: new wire b in {0,1}; b * (1 - b) = 0 -> A={b:1} B={0:1,b:-1} C={}
                bval = rng.below(2)
                wires.append(bval)
                wb = len(wires) - 1
                rows.append(({wb: 1}, {0: 1, wb: R_MOD - 1}, {}))
            else:
                A, B = lin(), lin()
                av = sum(k * wires[i] for i, k in A.items()) % R_MOD
                bv = sum(k * wires[i] for i, k in B.items()) % R_MOD
                wires.append((av * bv) % R_MOD)
                rows.append((A, B, {len(wires) - 1: 1}))
        self.rows = rows
        self.wires = wires
        self.nb_wires = len(wires)
        self.nb_constraints = nb_constraints

    def solve(self):
        """What gnark's solver hands the prove path: W and the per-row a, b, c values."""
        w = self.wires
        ev = lambda d: sum(k * w[i] for i, k in d.items()) % R_MOD
        a = [ev(r[0]) for r in self.rows]
        b = [ev(r[1]) for r in self.rows]
        c = [ev(r[2]) for r in self.rows]
        for x, y, z in zip(a, b, c):
            assert (x * y - z) % R_MOD == 0
        return w, a, b, c


class ToyTrapdoor:
    def __init__(self, seed: int):
        rng = SplitMix64(seed ^ 0x7A7A)
        self.tau, self.alpha, self.beta, self.gamma, self.delta = (rng.fr() for _ in range(5))


def lagrange_at(dom: Domain, tau: int):
    """L_i(tau) for the domain <w> of size n."""
    n = dom.n
    zt = (pow(tau, n, R_MOD) - 1) % R_MOD
    out = []
    wi = 1
    for _ in range(n):
        out.append((zt * wi * dom.card_inv * fr_inv((tau - wi) % R_MOD)) % R_MOD)
        wi = (wi * dom.gen) % R_MOD
    return out


def toy_setup(cs: ToyR1CS, td: ToyTrapdoor):
    """groth16.Setup by behaviour (mt.go:448): returns pk dict laid out as gnark keeps it
    (InfinityA/B masks, points-at-infinity filtered out, G1.Z bit-reversed, K private-only)
    plus the scalar 'exponents' of every pk element for the trapdoor check."""
    dom = Domain(cs.nb_constraints)
    n = dom.n
    L = lagrange_at(dom, td.tau)
    A = [0] * cs.nb_wires
    B = [0] * cs.nb_wires
    C = [0] * cs.nb_wires
    for i, (ra, rb, rc) in enumerate(cs.rows):
        for j, k in ra.items():
            A[j] = (A[j] + k * L[i]) % R_MOD
        for j, k in rb.items():
            B[j] = (B[j] + k * L[i]) % R_MOD
        for j, k in rc.items():
            C[j] = (C[j] + k * L[i]) % R_MOD
    dinv = fr_inv(td.delta)
    K = [((td.beta * A[j] + td.alpha * B[j] + C[j]) * dinv) % R_MOD for j in range(cs.nb_wires)]
    zt = ((pow(td.tau, n, R_MOD) - 1) * dinv) % R_MOD
    Z = [(zt * pow(td.tau, i, R_MOD)) % R_MOD for i in range(n)]
    inf_a = [x == 0 for x in A]
    inf_b = [x == 0 for x in B]
    g1 = lambda s: g1_mul(G1_GEN, s)
    g2 = lambda s: g2_mul(G2_GEN, s)
    pk = {
        "log_n": dom.log_n, "nb_wires": cs.nb_wires, "nb_public": cs.nb_public,
        "inf_a": inf_a, "inf_b": inf_b,
        "g1_a": [g1(x) for x in A if x != 0],
        "g1_b": [g1(x) for x in B if x != 0],
        "g2_b": [g2(x) for x in B if x != 0],
        "g1_k": [g1(K[j]) for j in range(cs.nb_public, cs.nb_wires)],
        "g1_z": bit_reverse_perm([g1(z) for z in Z]),
        "alpha1": g1(td.alpha), "beta1": g1(td.beta), "delta1": g1(td.delta),
        "beta2": g2(td.beta), "delta2": g2(td.delta),
    }
    exps = {"A": A, "B": B, "C": C, "K": K, "Z": Z}
    return pk, exps, dom


def toy_prove(cs: ToyR1CS, pk, dom: Domain, r: int, s: int, msm=msm_naive):
    """groth16.Prove after the solve, no BSB22 commitment."""
    w, a, b, c = cs.solve()
    h = compute_h(a, b, c, dom)
    wa = [w[j] for j in range(cs.nb_wires) if not pk["inf_a"][j]]
    wb = [w[j] for j in range(cs.nb_wires) if not pk["inf_b"][j]]
    wk = w[cs.nb_public:]
    kr = (-(r * s)) % R_MOD
    d1 = pk["delta1"]
    ar = g1_add(g1_add(msm(F1, pk["g1_a"], wa), pk["alpha1"]), g1_mul(d1, r))
    bs1 = g1_add(g1_add(msm(F1, pk["g1_b"], wb), pk["beta1"]), g1_mul(d1, s))
    krs = g1_add(msm(F1, pk["g1_k"], wk), msm(F1, pk["g1_z"][: dom.n - 1], h[: dom.n - 1]))
    krs = g1_add(krs, g1_mul(d1, kr))
    krs = g1_add(krs, g1_mul(ar, s))
    krs = g1_add(krs, g1_mul(bs1, r))
    bs = g2_add(g2_add(msm(F2, pk["g2_b"], wb), pk["beta2"]), g2_mul(pk["delta2"], s))
    return {"ar": ar, "bs": bs, "krs": krs, "h": h}


def trapdoor_check(cs: ToyR1CS, td: ToyTrapdoor, exps, proof, r: int, s: int) -> bool:
    """Groth16 verification equation in the exponent (no pairing needed):
    ar*bs == alpha*beta + sum_pub w_j*(beta A_j + alpha B_j + C_j) + krs*delta,
    with ar, bs, krs the discrete logs implied by (W, h, r, s)."""
    w = cs.wires
    A, B, K = exps["A"], exps["B"], exps["K"]
    ar = (td.alpha + sum(w[j] * A[j] for j in range(cs.nb_wires)) + r * td.delta) % R_MOD
    bs = (td.beta + sum(w[j] * B[j] for j in range(cs.nb_wires)) + s * td.delta) % R_MOD
    if proof["ar"] != g1_mul(G1_GEN, ar) or proof["bs"] != g2_mul(G2_GEN, bs):
        return False
    # krs is whatever the prover output; recover its dlog from the definition using h
    n = len(exps["Z"])
    h_nat = bit_reverse_perm(proof["h"])
    hz = sum(h_nat[i] * exps["Z"][i] for i in range(n - 1)) % R_MOD
    krs = (sum(w[j] * K[j] for j in range(cs.nb_public, cs.nb_wires)) + hz
           + s * ar + r * bs - r * s * td.delta) % R_MOD
    if proof["krs"] != g1_mul(G1_GEN, krs):
        return False
    pub = sum(w[j] * ((td.beta * A[j] + td.alpha * B[j] + exps["C"][j]) % R_MOD)
              for j in range(cs.nb_public)) % R_MOD
    return (ar * bs - td.alpha * td.beta - pub - krs * td.delta) % R_MOD == 0


# --------------------------------------------------------------------------- BSB22 Pedersen (gnark-crypto fr/pedersen by behaviour)
def pedersen_commit(basis, values):
    """ProvingKey.Commit: sum_i values[i] * Basis[i]  (ProveKnowledge: the same over BasisExpSigma)"""
    return msm_naive(F1, basis[: len(values)], values)


def pedersen_fold(points, challenge):
    """pedersen.Fold: sum_i challenge^i * points[i]"""
    acc, pw = None, 1
    for pt in points:
        acc = g1_add(acc, g1_mul(pt, pw))
        pw = (pw * challenge) % R_MOD
    return acc


# --------------------------------------------------------------------------- gnark-crypto point encoding
M_COMPRESSED_SMALLEST = 0b10 << 6
M_COMPRESSED_LARGEST = 0b11 << 6
M_COMPRESSED_INFINITY = 0b01 << 6
M_UNCOMPRESSED = 0b00 << 6


def _fp_lex_largest(y: int) -> bool:
    return y > (Q_MOD - 1) // 2


def g1_compress(P) -> bytes:
    if P is None:
        return bytes([M_COMPRESSED_INFINITY]) + bytes(31)
    x, y = P
    b = bytearray(x.to_bytes(32, "big"))
    b[0] |= M_COMPRESSED_LARGEST if _fp_lex_largest(y) else M_COMPRESSED_SMALLEST
    return bytes(b)


def g2_compress(P) -> bytes:
    if P is None:
        return bytes([M_COMPRESSED_INFINITY]) + bytes(63)
    (x0, x1), (y0, y1) = P
    largest = _fp_lex_largest(y1) if y1 != 0 else _fp_lex_largest(y0)
    b = bytearray(x1.to_bytes(32, "big") + x0.to_bytes(32, "big"))
    b[0] |= M_COMPRESSED_LARGEST if largest else M_COMPRESSED_SMALLEST
    return bytes(b)


def g1_uncompressed(P) -> bytes:
    if P is None:
        return bytes(64)  # gnark-crypto: mUncompressedInfinity = 0b01<<6 ... handled by caller if needed
    return P[0].to_bytes(32, "big") + P[1].to_bytes(32, "big")


def proof_bytes(proof, commitments=(), commitment_pok=None) -> bytes:
    """Proof.WriteTo order: Ar, Bs, Krs, u32-BE len + Commitments, CommitmentPok."""
    out = g1_compress(proof["ar"]) + g2_compress(proof["bs"]) + g1_compress(proof["krs"])
    out += len(commitments).to_bytes(4, "big")
    for cpt in commitments:
        out += g1_compress(cpt)
    out += g1_compress(commitment_pok)
    return out


# --------------------------------------------------------------------------- synthetic workload
def synth_scalar(rng: SplitMix64, dist: str) -> int:
    if dist == "uniform":
        return rng.fr()
    u = rng.below(100)  # "whir" mixture: 45% {0,1}, 25% bytes, 5% 64-bit, 25% uniform
    if u < 45:
        return rng.below(2)
    if u < 70:
        return rng.below(256)
    if u < 75:
        return rng.next()
    return rng.fr()


def synth_g1_point(rng: SplitMix64):
    """try-and-increment: x = PRNG mod q, y = (x^3+3)^((q+1)/4); smaller root kept if rng bit 0."""
    x = 0
    for k in range(4):
        x |= rng.next() << (64 * k)
    x %= Q_MOD
    while True:
        rhs = (x * x * x + G1_B) % Q_MOD
        y = pow(rhs, (Q_MOD + 1) // 4, Q_MOD)
        if (y * y) % Q_MOD == rhs:
            return (x, y)
        x = (x + 1) % Q_MOD
