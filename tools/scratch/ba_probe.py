"""Batch-affine level 1 (MI_MSM_BA_ROUNDS): G1 MSM 2^23 uniform / whir, generic and fixed-base timing; result equality across settings is checked by the caller comparing the printed hashes."""
import os, sys, hashlib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
from gpu_common import load_binding
B = load_binding(); ctx = B.Context(0)
n = 1 << 23
pts = ctx.gen_g1(n, 11); su = ctx.gen_scalars(n, 12, 0); sw = ctx.gen_scalars(n, 13, 1)
for name, sc in (("uniform", su), ("whir", sw)):
    for rep in range(3):
        out = ctx.msm_g1_dev(pts.ptr, sc.ptr, n); st = ctx.stats()
    print("g1 2^23", name, "total %.2f ms accum %.2f ms" % (st["total_ms"], st["g1_accum_kernel_ms"]), "entries", st.get("g1_accum_entries"), "sha", hashlib.sha256(bytes(out)).hexdigest()[:12], flush=True)
ctx.close()
