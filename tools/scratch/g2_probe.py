import os, sys
sys.path.insert(0, "tests")
from gpu_common import load_binding
B = load_binding(); ctx = B.Context(0)
n = 1 << 22
pts = ctx.gen_g2(n, 31); sc = ctx.gen_scalars(n, 32, 0)
for limb29 in (1, 0, 1, 0):
    assert ctx.lib.mi_debug_set_msm_limb29(ctx.h, limb29) == 0
    for _ in range(3):
        ctx.msm_g2_dev(pts.ptr, sc.ptr, n); st = ctx.stats()
    print(f"G2 2^22 uniform pairs limb29={limb29}: total {st['total_ms']:.2f} ms", flush=True)
ctx.close()
