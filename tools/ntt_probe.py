"""computeH alone on the GPU at N = 2^log_n (default 23): the workload for rocprofv3 --pmc / --kernel-trace passes over the NTT
kernels, and (with `sweep`) a sweep of the plan knobs.  usage: python3 tools/ntt_probe.py [log_n] [reps] [sweep]"""
import json
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from gpu_common import load_binding

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 23
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
sweep = len(sys.argv) > 3 and sys.argv[3] == "sweep"
B = load_binding()
ctx = B.Context(0)
N = 1 << log_n
nc = N - 100
a = ctx.gen_scalars(nc, 9, 1); b = ctx.gen_scalars(nc, 10, 0); c = ctx.alloc(32 * nc)
ctx.field_op_dev(0, 2, c.ptr, a.ptr, b.ptr, nc)
h = ctx.alloc(32 * N)


def run():
    best = None
    for _ in range(reps):
        ctx.compute_h_dev(log_n, a.ptr, b.ptr, c.ptr, nc, h.ptr)
        ms = ctx.stats()["compute_h_ms"]
        best = ms if best is None else min(best, ms)
    return best


print(f"computeH N=2^{log_n}: best of {reps}: {run():.3f} ms", flush=True)
if sweep:
    out = []
    ref = h.download((N, 4)).copy()
    cfgs26 = [(9, 9, 7, 256, 1), (10, 9, 9, 256, 1), (11, 9, 9, 256, 1), (10, 10, 8, 256, 1), (11, 11, 8, 256, 1), (10, 8, 9, 256, 1), (11, 8, 9, 256, 1),
              (10, 9, 8, 256, 1), (11, 9, 8, 256, 1), (11, 10, 8, 256, 1), (9, 9, 9, 256, 1), (9, 8, 9, 256, 1), (9, 9, 8, 256, 1)]
    for (log_e, mc, ms_, thr, wave) in cfgs26 if log_n >= 25 else [(9, 9, 7, 256, 1), (9, 9, 7, 256, 0), (9, 9, 7, 128, 1), (10, 9, 7, 256, 1), (10, 9, 7, 512, 1), (10, 10, 7, 256, 1),
                                        (9, 7, 8, 256, 1), (10, 7, 8, 256, 1), (10, 8, 8, 256, 1), (11, 9, 7, 256, 1), (11, 11, 7, 256, 1), (8, 8, 8, 128, 1),
                                        (9, 8, 8, 256, 1), (8, 7, 8, 128, 1), (8, 8, 8, 256, 1)]:
        assert ctx.lib.mi_debug_set_ntt_plan(ctx.h, log_e, mc, ms_) == 0 and ctx.lib.mi_debug_set_ntt_threads(ctx.h, thr) == 0
        assert ctx.lib.mi_debug_set_ntt_wave_stages(ctx.h, wave, 12) == 0
        t = run()
        ok = bool((h.download((N, 4)) == ref).all())
        out.append({"log_e": log_e, "max_contig": mc, "max_strided": ms_, "threads": thr, "wave": wave, "compute_h_ms": t, "same_h": ok})
        print(out[-1], flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open("gpurun_out/r2_ntt_sweep.json", "w"), indent=1)
ctx.close()
