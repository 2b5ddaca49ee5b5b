"""GPU probes: VALU issue rates and modular-multiply throughput (writes gpurun_out/probe.json)."""
import json
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from gpu_common import load_binding

B = load_binding()
ctx = B.Context(0)
out = {}
nthr, iters = 256 * 2048, 4096
names = {0: "mad_u64_u32", 1: "mul_lo+mul_hi", 2: "fma_f64", 3: "mul24+add", 4: "add_u64"}
for kind, name in names.items():
    ctx.bench_valu(kind, nthr, 64)
    ms = min(ctx.bench_valu(kind, nthr, iters) for _ in range(3))
    ops = nthr * iters * 8
    out[name] = {"ms": ms, "Gops_per_s": ops / ms / 1e6}
    print(name, out[name], flush=True)
for field, name in ((0, "modmul_fr"), (1, "modmul_fp")):
    for nt in (256 * 256, 256 * 1024, 256 * 2048, 256 * 4096):
        ctx.bench_modmul(field, nt, 8)
        ms = min(ctx.bench_modmul(field, nt, 512) for _ in range(3))
        out[f"{name}_{nt}"] = {"ms": ms, "Gmul_per_s": nt * 512 * 2 / ms / 1e6}
        print(name, nt, out[f"{name}_{nt}"], flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/probe.json", "w"), indent=1)
ctx.close()
