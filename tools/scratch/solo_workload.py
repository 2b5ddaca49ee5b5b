"""Solo kernels for a clean per-kernel profile: G1 MSM 2^23 (uniform, whir), G2 MSM 2^22 (whir), computeH 2^23."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from gpu_common import load_binding

B = load_binding()
ctx = B.Context(0)
n = 1 << 23
pts = ctx.gen_g1(n, 11); su = ctx.gen_scalars(n, 12, 0); sw = ctx.gen_scalars(n, 13, 1)
for rep in range(2):
    ctx.msm_g1_dev(pts.ptr, su.ptr, n); print("g1 uniform", ctx.stats()["total_ms"], ctx.stats()["g1_accum_kernel_ms"], flush=True)
    ctx.msm_g1_dev(pts.ptr, sw.ptr, n); print("g1 whir", ctx.stats()["total_ms"], ctx.stats()["g1_accum_kernel_ms"], flush=True)
p2 = ctx.gen_g2(n // 2, 14)
for rep in range(2):
    ctx.msm_g2_dev(p2.ptr, sw.ptr, n // 2); print("g2 whir 2^22", ctx.stats()["total_ms"], flush=True)
h = ctx.alloc(32 * n)
for rep in range(2):
    ctx.compute_h_dev(23, su.ptr, sw.ptr, pts.ptr, n - 100, h.ptr); print("compute_h", ctx.stats()["compute_h_ms"], flush=True)
ctx.close()
