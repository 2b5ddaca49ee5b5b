"""One generic (arbitrary bases, c = 16) G1 MSM of 2^log_n pairs alone on the GPU, with the two-pass and the one-pass sort.
usage: python3 tools/msm_generic_probe.py [log_n] [dist]   (under rocprofv3 --kernel-trace / --pmc for the per-kernel split)"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from gpu_common import load_binding

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 23
dist = int(sys.argv[2]) if len(sys.argv) > 2 else 0
B = load_binding(); ctx = B.Context(0)
n = 1 << log_n
pts = ctx.gen_g1(n, 31); sc = ctx.gen_scalars(n, 32, dist)
for one_pass, limb29 in ((1, 1), (0, 1), (0, 0), (0, 1), (0, 0)):
    assert ctx.lib.mi_debug_set_msm_one_pass_sort(ctx.h, one_pass) == 0 and ctx.lib.mi_debug_set_msm_limb29(ctx.h, limb29) == 0
    for _ in range(3):
        out = ctx.msm_g1_dev(pts.ptr, sc.ptr, n); st = ctx.stats()
    print(f"one_pass_sort={one_pass} limb29={limb29}: 2^{log_n} pairs dist {dist}: total {st['total_ms']:.2f} ms, accumulate {st['g1_accum_kernel_ms']:.2f} ms "
          f"({st['g1_accum_entries'] / st['g1_accum_kernel_ms'] / 1e6:.2f} G adds/s), {n / st['total_ms'] / 1e3:.1f} M pts/s", flush=True)
ctx.close()
