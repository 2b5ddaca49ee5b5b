"""Small fixed workload for rocprofv3 --pmc passes: one G1 MSM (2^23 uniform pairs) and one size-2^23 NTT."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from gpu_common import load_binding

B = load_binding()
ctx = B.Context(0)
n = 1 << 23
pts = ctx.gen_g1(n, 11); s = ctx.gen_scalars(n, 12, 0)
ctx.msm_g1_dev(pts.ptr, s.ptr, n)
print("msm stats", ctx.stats())
ctx.ntt_dev(s.ptr, 23, 1)
print("ntt stats", ctx.stats())
ctx.close()
