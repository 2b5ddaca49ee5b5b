"""Replays the value / limb bounds of the 29-bit-limb mixed addition (gnark-whir_amd/csrc/curve29.cuh) with worst-case interval
arithmetic and checks every precondition of the primitives in field29.cuh.  V = value bound in multiples of p, L = limb bound in
bits (limbs 0..7).  Run: python tools/f29_bounds.py"""
import math

WEAK = math.log2(2 ** 29 + 8)


class B:
    def __init__(self, V, L, name):
        self.V, self.L, self.name = V, L, name


def mul(x, y, name):
    assert x.L + y.L <= 60.0 + 1e-6, (name, x.L, y.L)          # 9 * 2^60 + 9 * 2^58 + carry < 2^64
    V = x.V * y.V / 128 + 1
    assert V < 64, name
    return B(V, 29, name)


def mul2(a, b, c, d, name):
    assert a.L + b.L <= 59.0 + 1e-6 and c.L + d.L <= 59.0 + 1e-6, name   # 18 * 2^59 + 9 * 2^58 < 2^64
    return B((a.V * b.V + c.V * d.V) / 128 + 1, 29, name)


def add(x, y, name):
    L = math.log2(2 ** x.L + 2 ** y.L)
    assert L <= 32, name
    return B(x.V + y.V, L, name)


def sub(x, y, K, name):
    assert y.L <= WEAK + 1e-9 and y.V < K - 0.01, (name, y.L, y.V)      # borrowed K p: limbs >= 2^30 - 2, top limb = (K p >> 232) - 2
    L = math.log2(2 ** x.L + 2 ** 30.59)
    assert L <= 32, name
    return B(x.V + K, L, name)


def wnorm(x, name=None):
    assert x.L <= 32
    return B(x.V, WEAK, name or x.name)


def round_(X, Y, ZZ, ZZZ):
    x2 = B(1, 29, "x2"); y2 = B(1, 29, "y2")
    U2 = mul(x2, ZZ, "U2"); S2 = mul(y2, ZZZ, "S2")
    P = wnorm(sub(U2, X, 8, "P")); R = wnorm(sub(S2, Y, 8, "R"))
    PP = mul(P, P, "PP (f29_sqr: same columns as the product)"); PPP = mul(P, PP, "PPP"); Q = mul(X, PP, "Q")
    T = wnorm(add(add(PPP, Q, "T"), Q, "T"))
    RR = mul(R, R, "RR (f29_sqr)")
    X3 = wnorm(sub(RR, T, 4, "X3"))
    D = wnorm(sub(Q, X3, 8, "D"))
    nY = wnorm(sub(B(0, 0, "0"), Y, 8, "nY"))
    Y3 = mul2(R, D, nY, PPP, "Y3")
    return X3, Y3, mul(ZZ, PP, "ZZ3"), mul(ZZZ, PPP, "ZZZ3")


state = (B(1, 29, "X"), B(1, 29, "Y"), B(1.01, 29, "ZZ"), B(1.01, 29, "ZZZ"))   # right after the first point
for it in range(12):
    state = round_(*state)
print("fixed point of the accumulator bounds:", ", ".join(f"{b.name} V < {b.V:.2f} (L {b.L:.2f})" for b in state))
assert state[0].V < 5.7 and state[1].V < 1.8 and state[2].V < 1.04 and state[3].V < 1.04


def add_round(a, b):
    """g1x29_add: a = running sum, b = a loaded partial sum (normalised, X < 4.02 after g1x29_store_rp's conditional subtraction)"""
    (Xa, Ya, ZZa, ZZZa), (Xb, Yb, ZZb, ZZZb) = a, b
    U1 = mul(Xa, ZZb, "U1"); U2 = mul(Xb, ZZa, "U2"); S1 = mul(Ya, ZZZb, "S1"); S2 = mul(Yb, ZZZa, "S2")
    P = wnorm(sub(U2, U1, 2, "P")); R = wnorm(sub(S2, S1, 2, "R"))
    PP = mul(P, P, "PP"); PPP = mul(P, PP, "PPP"); Q = mul(U1, PP, "Q")
    T = wnorm(add(add(PPP, Q, "T"), Q, "T")); RR = mul(R, R, "RR")
    X3 = wnorm(sub(RR, T, 4, "X3")); D = wnorm(sub(Q, X3, 8, "D")); nS1 = wnorm(sub(B(0, 0, "0"), S1, 2, "nS1"))
    return X3, mul2(R, D, nS1, PPP, "Y3"), mul(mul(ZZa, ZZb, "ZZab"), PP, "ZZ3"), mul(mul(ZZZa, ZZZb, "ZZZab"), PPP, "ZZZ3")


stored = (B(4.02, 29, "Xb"), B(state[1].V, 29, "Yb"), B(state[2].V, 29, "ZZb"), B(state[3].V, 29, "ZZZb"))   # what g1x29_load_rp returns at worst
run = (B(5.7, WEAK, "Xa"), stored[1], stored[2], stored[3])
for it in range(12):
    run = add_round(run, stored)
    assert run[0].V < 5.7 and run[1].V < 1.8 and run[2].V < 1.04 and run[3].V < 1.04, [b.V for b in run]
print("g1x29_add keeps the accumulator invariant:", ", ".join(f"{b.name} V < {b.V:.2f}" for b in run))
print("all preconditions hold; to_std needs V <= 128:", max(b.V for b in state), "ok")
