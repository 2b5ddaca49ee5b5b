"""Writes tests/golden/* from the definitional python oracle (oracle/pyref.py).

Run:  python oracle/gen_golden.py        (about a minute; deterministic)
The fixtures are DATA (inputs + expected outputs); both oracles and the HIP path are tested
against them.  No reference code is involved: the reference has no vectors for this path and
cannot run here, so these pin the build against drift, not against gnark.
"""
import json
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", "tests"))
import pyref as P  # noqa: E402
import cref  # noqa: E402
from helpers import fr_arr, fr_vals, g1_arr, g1_pts, g2_arr, g2_pts, toy_pk_arrays  # noqa: E402

OUT = os.path.join(HERE, "..", "tests", "golden")
os.makedirs(OUT, exist_ok=True)
hx = lambda v: "%x" % v


def gen_ntt():
    rng = P.SplitMix64(0x57484952)
    cases = []
    for logn in (1, 3, 6):
        n = 1 << logn
        dom = P.Domain(n)
        a = [rng.fr() for _ in range(n)]
        for flags in range(8):
            f = P.fft_inverse if flags & 1 else P.fft
            out = f(dom, a, P.DIT if flags & 4 else P.DIF, coset=bool(flags & 2))
            cases.append({"log_n": logn, "flags": flags, "in": [hx(v) for v in a], "out": [hx(v) for v in out]})
    n = 1024
    dom = P.Domain(n)
    a = [rng.fr() for _ in range(n)]
    for flags in (1, 6, 3):  # the three transforms computeH uses
        f = P.fft_inverse if flags & 1 else P.fft
        out = f(dom, a, P.DIT if flags & 4 else P.DIF, coset=bool(flags & 2))
        cases.append({"log_n": 10, "flags": flags, "in": [hx(v) for v in a], "out": [hx(v) for v in out]})
    json.dump({"cases": cases}, open(os.path.join(OUT, "ntt.json"), "w"))


def gen_msm():
    rng = P.SplitMix64(0x57484953)
    g1 = []
    for n, dist in ((1, "uniform"), (2, "whir"), (255, "uniform"), (255, "whir")):
        pts = [P.synth_g1_point(rng) for _ in range(n)]
        sc = [P.synth_scalar(rng, dist) for _ in range(n)]
        if n > 2:
            pts[7] = None
            sc[0], sc[1], sc[2], sc[3] = 0, 1, P.R_MOD - 1, P.R_MOD - 2
            pts[9] = pts[8]                      # repeated point -> doubling inside a bucket
            sc[9] = sc[8]
            pts[11] = P.g1_neg(pts[10]); sc[11] = sc[10]   # P + (-P) inside a bucket
        out = P.msm_pippenger(P.F1, pts, sc, 6) if n > 40 else P.msm_naive(P.F1, pts, sc)
        g1.append({"points": [None if p is None else [hx(p[0]), hx(p[1])] for p in pts],
                   "scalars": [hx(s) for s in sc], "out": None if out is None else [hx(out[0]), hx(out[1])]})
    g2 = []
    for n in (1, 16):
        pts = [P.g2_mul(P.G2_GEN, 1 + rng.next()) for _ in range(n)]
        sc = [P.synth_scalar(rng, "whir" if n > 1 else "uniform") for _ in range(n)]
        if n > 1:
            sc[0], sc[1], sc[2] = 0, 1, P.R_MOD - 1
            pts[3] = None
        out = P.msm_naive(P.F2, pts, sc)
        flat = lambda p: None if p is None else [hx(p[0][0]), hx(p[0][1]), hx(p[1][0]), hx(p[1][1])]
        g2.append({"points": [flat(p) for p in pts], "scalars": [hx(s) for s in sc], "out": flat(out)})
    json.dump({"g1": g1, "g2": g2}, open(os.path.join(OUT, "msm.json"), "w"))
    # n = 4096: inputs from the seeded C generators, expected value from the python definition
    for dist, name in ((0, "uniform"), (1, "whir")):
        pts = cref.gen_g1(4096, 0x4096 + dist); sc = cref.gen_scalars(4096, 0x1000 + dist, dist)
        out = P.msm_pippenger(P.F1, g1_pts(pts), fr_vals(sc), 8)
        np.savez_compressed(os.path.join(OUT, f"msm_g1_4096_{name}.npz"), points=pts, scalars=sc, out=g1_arr([out])[0])
    pts = cref.gen_g2(512, 0x512); sc = cref.gen_scalars(512, 0x513, 1)
    out = P.msm_pippenger(P.F2, g2_pts(pts), fr_vals(sc), 6)
    np.savez_compressed(os.path.join(OUT, "msm_g2_512_whir.npz"), points=pts, scalars=sc, out=g2_arr([out])[0])


def gen_prove():
    nc, npub, seed = 1000, 9, 0x57484954
    cs = P.ToyR1CS(nc, npub, seed); td = P.ToyTrapdoor(seed)
    pk, exps, dom = P.toy_setup(cs, td)
    rng = P.SplitMix64(seed + 1); r, s = rng.fr(), rng.fr()
    pr = P.toy_prove(cs, pk, dom, r, s, msm=lambda F, p, sc: P.msm_pippenger(F, p, sc, 6))
    assert P.trapdoor_check(cs, td, exps, pr, r, s)
    json.dump({"nb_constraints": nc, "nb_public": npub, "seed": seed, "r": hx(r), "s": hx(s),
               "proof_bytes": P.proof_bytes(pr).hex()}, open(os.path.join(OUT, "prove.json"), "w"))
    w, a, b, c = cs.solve()
    arrs = toy_pk_arrays(pk)
    np.savez_compressed(os.path.join(OUT, "prove_toy1000.npz"), W=fr_arr(w), a=fr_arr(a), b=fr_arr(b), c=fr_arr(c),
                        r=fr_arr([r])[0], s=fr_arr([s])[0], h=fr_arr(pr["h"]),
                        ar=g1_arr([pr["ar"]])[0], bs=g2_arr([pr["bs"]])[0], krs=g1_arr([pr["krs"]])[0],
                        proof_bytes=np.frombuffer(P.proof_bytes(pr), dtype=np.uint8), **arrs)


if __name__ == "__main__":
    gen_ntt(); print("ntt ok")
    gen_msm(); print("msm ok")
    gen_prove(); print("prove ok")
