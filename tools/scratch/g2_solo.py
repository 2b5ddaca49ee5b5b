"""G2 MSM alone, 2^22 pairs, WHIR mix and uniform: total time and the proof-independent result hash (A/B of two library builds through MI355X_GROTH16_LIB)."""
import os, sys, hashlib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
from gpu_common import load_binding
B = load_binding(); ctx = B.Context(0)
n = 1 << 22
p2 = ctx.gen_g2(n, 14); su = ctx.gen_scalars(n, 12, 0); sw = ctx.gen_scalars(n, 13, 1)
for name, sc in (("uniform", su), ("whir", sw)):
    ts = []
    for rep in range(5):
        out = ctx.msm_g2_dev(p2.ptr, sc.ptr, n); ts.append(ctx.stats()["total_ms"])
    print("g2 2^22", name, "total ms min %.2f median %.2f" % (min(ts), sorted(ts)[2]), "sha", hashlib.sha256(bytes(out)).hexdigest()[:12], flush=True)
ctx.close()
