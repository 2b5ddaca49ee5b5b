"""GPU tuning sweep over the debug knobs (NTT tile/threads/plan, MSM c/L1/L2/seg/G). Writes gpurun_out/tune.json."""
import ctypes as C
import json
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from gpu_common import load_binding

B = load_binding()
ctx = B.Context(0)
lib = ctx.lib
out = {}
which = sys.argv[1:] or ["ntt", "msm"]
log_n = 23
n = 1 << log_n
if "ntt" in which:
    x = ctx.gen_scalars(n, 1, 0)
    for threads in (256,):
        for (log_e, mc, ms) in ((9, 9, 7), (12, 11, 12), (12, 12, 11), (11, 11, 11), (10, 10, 10), (11, 9, 7)):
            assert lib.mi_debug_set_ntt_threads(ctx.h, threads) == 0
            if lib.mi_debug_set_ntt_plan(ctx.h, log_e, mc, ms) != 0:
                continue
            best = 1e9
            for flags in (1, 6):
                for _ in range(3):
                    ctx.ntt_dev(x.ptr, log_n, flags)
                    best = min(best, ctx.stats()["ntt_kernel_ms"])
            out[f"ntt_t{threads}_e{log_e}_c{mc}_s{ms}"] = best
            print("ntt", threads, log_e, mc, ms, best, flush=True)
    lib.mi_debug_set_ntt_plan(ctx.h, 9, 9, 7); lib.mi_debug_set_ntt_threads(ctx.h, 256)
    x.free()
if "msm" in which:
    pts = ctx.gen_g1(n, 11)
    for dist in (0, 1):
        s = ctx.gen_scalars(n, 12, dist)
        for (c, L1, L2, seg, G) in ((0, 0, 0, 0, 0), (16, 16, 16, 8, 64), (16, 64, 16, 8, 64), (16, 32, 32, 8, 64), (16, 32, 8, 8, 64),
                                    (16, 32, 16, 4, 64), (16, 32, 16, 16, 64), (16, 32, 4, 8, 64), (16, 32, 4, 4, 64), (16, 16, 8, 8, 64), (16, 16, 4, 4, 64)):
            assert lib.mi_debug_set_msm_plan(ctx.h, c, L1, L2, seg, G) == 0
            best, acc = 1e9, 1e9
            for _ in range(2):
                ctx.msm_g1_dev(pts.ptr, s.ptr, n)
                st = ctx.stats(); best = min(best, st["total_ms"]); acc = min(acc, st["g1_accum_kernel_ms"])
            out[f"msm_d{dist}_c{c}_L{L1}_{L2}_seg{seg}_G{G}"] = [best, acc]
            print("msm", dist, c, L1, L2, seg, G, best, acc, flush=True)
        s.free()
    lib.mi_debug_set_msm_plan(ctx.h, 0, 0, 0, 0, 0)
if "g2" in which:
    p2 = ctx.gen_g2(n // 2, 13); s2 = ctx.gen_scalars(n // 2, 14, 1)
    for (c, L1, L2, seg, G) in ((0, 0, 0, 0, 0), (16, 32, 8, 8, 64), (16, 32, 4, 8, 64), (16, 32, 4, 4, 64), (16, 32, 8, 4, 64), (16, 16, 8, 8, 64), (16, 16, 4, 4, 64), (16, 32, 2, 4, 64)):
        assert lib.mi_debug_set_msm_plan(ctx.h, c, L1, L2, seg, G) == 0
        best = 1e9
        for _ in range(2):
            ctx.msm_g2_dev(p2.ptr, s2.ptr, n // 2)
            best = min(best, ctx.stats()["total_ms"])
        out[f"g2_c{c}_L{L1}_{L2}_seg{seg}"] = best
        print("msm_g2 whir 2^22", c, L1, L2, seg, G, best, flush=True)
    lib.mi_debug_set_msm_plan(ctx.h, 0, 0, 0, 0, 0)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/tune.json", "w"), indent=1)
ctx.close()
