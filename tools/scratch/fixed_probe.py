"""Solo fixed-base MSM vs the generic one at 2^23 pairs.  usage: fixed_probe.py [c] [gbits] [chunk]"""
import sys
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from gpu_common import load_binding
B = load_binding(); ctx = B.Context(0)
n, c = 1 << 23, int(sys.argv[1]) if len(sys.argv) > 1 else 20
gbits = int(sys.argv[2]) if len(sys.argv) > 2 else 0
chunk = int(sys.argv[3]) if len(sys.argv) > 3 else 0
assert ctx.lib.mi_debug_set_msm_group_bits(ctx.h, gbits) == 0 and ctx.lib.mi_debug_set_msm_chunk(ctx.h, chunk) == 0
pts = ctx.gen_g1(n, 31)
pre = ctx.msm_precompute(pts.ptr, n, c)
for dist in (0, 1):
    sc = ctx.gen_scalars(n, 32, dist)
    for rep in range(3):
        ctx.msm_g1_dev(pts.ptr, sc.ptr, n); tg = ctx.stats()
        ctx.msm_fixed_dev(pre.ptr, sc.ptr, n, c); tf = ctx.stats()
    print(f"dist {dist}: generic {tg['total_ms']:.2f} ms (accum {tg['g1_accum_kernel_ms']:.2f}, {tg['g1_accum_entries']} entries); fixed c={c} gbits={gbits} chunk={chunk} {tf['total_ms']:.2f} ms (accum {tf['g1_accum_kernel_ms']:.2f}, {tf['g1_accum_entries']} entries)", flush=True)
ctx.close()
