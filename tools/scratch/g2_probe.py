"""One G2 MSM alone on the GPU (mi_msm_g2_dev, generic path): time and parity of two builds.   usage: python3 tools/scratch/g2_probe.py [log_n] [dist]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import numpy as np
from gpu_common import load_binding
B = load_binding()
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 22
dist = int(sys.argv[2]) if len(sys.argv) > 2 else 0
n = 1 << log_n
ctx = B.Context(0)
pts = ctx.gen_g2(n, 77); sc = ctx.gen_scalars(n, 78, dist)
ctx.sync()
out = ctx.msm_g2_dev(pts.ptr, sc.ptr, n)
ms = []
for _ in range(4):
    t0 = time.perf_counter(); out2 = ctx.msm_g2_dev(pts.ptr, sc.ptr, n); ms.append((time.perf_counter() - t0) * 1e3)
    assert np.array_equal(out, out2)
print(os.environ.get("MI355X_GROTH16_LIB", "in-tree"), "G2 MSM 2^%d dist %d: %.2f ms (min of 4), result word0 %x" % (log_n, dist, min(ms), int(out[0])), flush=True)
ctx.close()
