"""How often does a mid-solve-sized call hiccup?  200 Pedersen commits at n = 2^18, values in pageable vs pinned host memory."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
from gpu_common import load_binding
import cref, torch
B = load_binding(); ctx = B.Context(0)
n = 1 << 18
basis = cref.gen_g1(n, 1); vals = cref.gen_scalars(n, 2, 1)
pk = ctx.pedersen_pk_load(basis, basis)
pinned = torch.from_numpy(vals.view(np.int64)).pin_memory().numpy().view(np.uint64)
dv = ctx.to_dev(vals); db = ctx.to_dev(basis)
for name, arr in (("pageable", vals), ("pinned", pinned)):
    ts = []
    for _ in range(200):
        t = time.perf_counter(); ctx.pedersen_commit(pk, arr); ts.append((time.perf_counter() - t) * 1e3)
    ts = np.array(ts)
    print(f"{name}: median {np.median(ts):.2f} ms, p99 {np.percentile(ts, 99):.2f}, max {ts.max():.2f}, calls > 10 ms: {(ts > 10).sum()}", flush=True)
ts = []
for _ in range(200):
    t = time.perf_counter(); ctx.msm_g1_dev(db.ptr, dv.ptr, n); ts.append((time.perf_counter() - t) * 1e3)
ts = np.array(ts)
print(f"device-resident msm: median {np.median(ts):.2f} ms, p99 {np.percentile(ts, 99):.2f}, max {ts.max():.2f}, calls > 10 ms: {(ts > 10).sum()}", flush=True)
