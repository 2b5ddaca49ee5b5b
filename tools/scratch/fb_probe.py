"""Fixed-base batch scalar multiplication timing: 2^log_n scalars against one G1 base and 2^(log_n - 2) against one G2 base.
usage: python3 tools/scratch/fb_probe.py [log_n]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "oracle"))
import ctypes as C
import numpy as np
import cref
from gpu_common import load_binding
B = load_binding()
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 23
ctx = B.Context(0)
for g2 in (False, True):
    n = 1 << (log_n - 2 if g2 else log_n)
    sc = ctx.gen_scalars(n, 5, 0)
    base = (cref.gen_g2 if g2 else cref.gen_g1)(1, 7)[0]
    out = ctx.alloc((128 if g2 else 64) * n)
    f = ctx.lib.mi_batch_scalar_mul_g2_dev if g2 else ctx.lib.mi_batch_scalar_mul_g1_dev
    bp = np.ascontiguousarray(base)
    for rep in range(3):
        ctx.sync(); t0 = time.perf_counter()
        assert f(ctx.h, bp.ctypes.data_as(C.c_void_p), C.c_void_p(sc.ptr), C.c_size_t(n), C.c_void_p(out.ptr)) == 0
        ctx.sync(); dt = time.perf_counter() - t0
    got = out.download((n, 16 if g2 else 8))[:64]
    want = cref.batch_scalar_mul(base, sc.download((n, 4))[:64], g2=g2)
    print("G2" if g2 else "G1", "n = 2^%d" % (log_n - 2 if g2 else log_n), "%.2f ms" % (dt * 1e3), "%.3e pts/s" % (n / dt), "first 64 equal oracle:", bool(np.array_equal(got, want)), flush=True)
ctx.close()
