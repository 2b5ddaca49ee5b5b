"""One fixed-base MSM alone on the GPU, shaped like one of the three sorts of a proof at N = 2^23:
   ak: n = 8.39 M scalars, WHIR mix, c = 19   b: n = 4.19 M, WHIR mix, c = 18   z: n = 8.39 M, uniform, c = 20
usage: python3 tools/sort_probe.py ak|b|z [reps]   (run under rocprofv3 --kernel-trace for the per-kernel split)"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from gpu_common import load_binding

which = sys.argv[1] if len(sys.argv) > 1 else "z"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
n, dist, c = {"ak": (8387608, 1, 19), "b": (4194488, 1, 18), "z": (8388607, 0, 20)}[which]
B = load_binding(); ctx = B.Context(0)
pts = ctx.gen_g1(n, 31); pre = ctx.msm_precompute(pts.ptr, n, c); sc = ctx.gen_scalars(n, 32, dist)
for _ in range(reps):
    ctx.msm_fixed_dev(pre.ptr, sc.ptr, n, c); st = ctx.stats()
print(f"{which}: n={n} c={c} total {st['total_ms']:.2f} ms accum {st['g1_accum_kernel_ms']:.2f} ms entries {st['g1_accum_entries']}", flush=True)
ctx.close()
