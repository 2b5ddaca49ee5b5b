"""computeH at 2^log_n alone on the GPU, default plan (for rocprofv3 --pmc / --kernel-trace).  usage: python3 tools/scratch/compute_h_solo.py [log_n] [reps]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
from gpu_common import load_binding
B = load_binding()
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 23
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
N = 1 << log_n
ctx = B.Context(0)
a = ctx.gen_scalars(N - 100, 1, 1); b = ctx.gen_scalars(N - 100, 2, 0); c = ctx.gen_scalars(N - 100, 3, 0)
h = ctx.alloc(32 * N)
ms = []
for _ in range(reps):
    ctx.compute_h_dev(log_n, a.ptr, b.ptr, c.ptr, N - 100, h.ptr)
    ms.append(ctx.stats()["compute_h_ms"])
print("computeH ms:", " ".join("%.3f" % x for x in ms), flush=True)
ctx.close()
