"""Do an MSM's sort and tails hide behind ANOTHER MSM's level 1?  K contexts (own streams and workspaces, one shared table) run the Z-shaped
fixed-base MSM (n = 2^23 - 1, uniform, c = 20, 29-bit level 1) back to back, each from its own thread; ms per MSM of the whole set.
usage: python3 tools/scratch/overlap_probe2.py [reps] [shape z|ak|b]"""
import os
import sys
import threading
import time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
from gpu_common import load_binding

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
which = sys.argv[2] if len(sys.argv) > 2 else "z"
n, dist, c = {"ak": (8387608, 1, 19), "b": (4194488, 1, 18), "z": (8388607, 0, 20)}[which]
B = load_binding()
ctxs = [B.Context(0) for _ in range(4)]
for x in ctxs:
    assert x.lib.mi_debug_set_msm_limb29(x.h, 1) == 0
c0 = ctxs[0]
pts = c0.gen_g1(n, 31); pre = c0.msm_precompute(pts.ptr, n, c); sc = c0.gen_scalars(n, 32, dist); c0.sync()


def loop(x, k, out, i):
    acc = 0.0
    for _ in range(k):
        x.msm_fixed_dev(pre.ptr, sc.ptr, n, c); acc += x.stats()["g1_accum_kernel_ms"]
    out[i] = acc / k


for x in ctxs:
    loop(x, 2, {}, 0)
for k in (1, 2, 3, 4):
    out = {}
    th = [threading.Thread(target=loop, args=(ctxs[i], reps, out, i)) for i in range(k)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    dt = (time.perf_counter() - t0) * 1e3
    print(f"{which}: {k} context(s): {dt / (k * reps):.2f} ms per MSM; level-1 launch {sum(out.values()) / k:.2f} ms", flush=True)
for x in ctxs[1:]:
    x.close()
c0.close()
