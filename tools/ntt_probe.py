"""computeH alone on the GPU at N = 2^log_n (default 23), a few repetitions: the workload for rocprofv3 --pmc / --kernel-trace
passes over k_ntt_pass.  usage: python3 tools/ntt_probe.py [log_n] [reps]"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from gpu_common import load_binding

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 23
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
B = load_binding()
ctx = B.Context(0)
N = 1 << log_n
nc = N - 100
a = ctx.gen_scalars(nc, 9, 1); b = ctx.gen_scalars(nc, 10, 0); c = ctx.alloc(32 * nc)
ctx.field_op_dev(0, 2, c.ptr, a.ptr, b.ptr, nc)
h = ctx.alloc(32 * N)
best = None
for _ in range(reps):
    ctx.compute_h_dev(log_n, a.ptr, b.ptr, c.ptr, nc, h.ptr)
    ms = ctx.stats()["compute_h_ms"]
    best = ms if best is None else min(best, ms)
print(f"computeH N=2^{log_n}: best of {reps}: {best:.3f} ms", flush=True)
ctx.close()
