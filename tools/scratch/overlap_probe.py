"""Does memory-bound work hide behind the level-1 accumulate?  One fixed-base Z-shaped MSM (n = 2^23 - 1, uniform, c = 20, 29-bit level 1)
repeated on a context while torch streams device-to-device copies (1 GiB each) on its own stream: both alone, then together.
usage: python3 tools/scratch/overlap_probe.py [reps]"""
import os
import sys
import threading
import time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import torch
from gpu_common import load_binding

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n, c = 8388607, 20
B = load_binding(); ctx = B.Context(0)
assert ctx.lib.mi_debug_set_msm_limb29(ctx.h, 1) == 0
pts = ctx.gen_g1(n, 31); pre = ctx.msm_precompute(pts.ptr, n, c); sc = ctx.gen_scalars(n, 32, 0)
src = torch.empty(1 << 28, dtype=torch.int32, device="cuda"); dst = torch.empty_like(src)   # 1 GiB each
side = torch.cuda.Stream()


def msms(k):
    t0 = time.perf_counter()
    acc = 0.0
    for _ in range(k):
        ctx.msm_fixed_dev(pre.ptr, sc.ptr, n, c); acc += ctx.stats()["g1_accum_kernel_ms"]
    return (time.perf_counter() - t0) * 1e3 / k, acc / k


def copies(stop):
    k = 0
    t0 = time.perf_counter()
    with torch.cuda.stream(side):
        while not stop.is_set():
            for _ in range(4):
                dst.copy_(src, non_blocking=True)
            side.synchronize(); k += 4
    dt = time.perf_counter() - t0
    return k, k * 2 * src.numel() * 4 / dt / 1e12   # read + write


msms(2)
alone_ms, alone_acc = msms(reps)
stop = threading.Event(); out = {}
th = threading.Thread(target=lambda: out.update(c=copies(stop))); th.start(); time.sleep(0.5); stop.set(); th.join()
print(f"alone: MSM {alone_ms:.2f} ms (level 1 {alone_acc:.2f} ms); copies {out['c'][1]:.2f} TB/s (read + write)", flush=True)
stop = threading.Event(); out = {}
th = threading.Thread(target=lambda: out.update(c=copies(stop))); th.start(); time.sleep(0.05)
both_ms, both_acc = msms(reps)
stop.set(); th.join()
print(f"together: MSM {both_ms:.2f} ms (level 1 {both_acc:.2f} ms); copies {out['c'][1]:.2f} TB/s", flush=True)
ctx.close()
