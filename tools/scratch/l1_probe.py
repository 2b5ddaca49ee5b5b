import os, sys
sys.path.insert(0, "tests")
from gpu_common import load_binding
B = load_binding(); ctx = B.Context(0)
n = 1 << 23
pts = ctx.gen_g1(n, 31); sc = ctx.gen_scalars(n, 32, 0)
for limb29 in (1, 0):
    for L1 in (16, 32, 64, 8):
        assert ctx.lib.mi_debug_set_msm_limb29(ctx.h, limb29) == 0 and ctx.lib.mi_debug_set_msm_plan(ctx.h, 0, L1, 8, 0, 0) == 0
        for _ in range(3):
            ctx.msm_g1_dev(pts.ptr, sc.ptr, n); st = ctx.stats()
        print(f"limb29={limb29} L1={L1}: total {st['total_ms']:.2f} ms accumulate {st['g1_accum_kernel_ms']:.2f} ms ({st['g1_accum_entries'] / st['g1_accum_kernel_ms'] / 1e6:.2f} G adds/s)", flush=True)
ctx.close()
