"""computeH over the ranks against computeH on one context, SAME GPU (a single-process group that names device 0 `world` times: the
all-to-alls are device-to-device copies): what the four-step form costs in arithmetic and passes -- the factor the projection of
DESIGN.md section 6 must carry -- not a scaling measurement.   python3 tools/sharded_h_probe.py [log_n] [world]"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

B = bench._binding()
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 23
world = int(sys.argv[2]) if len(sys.argv) > 2 else 2
N = 1 << log_n
M = N // world
n = N - 100
g = B.Group([0] * world)
c0 = g.ctx(0)
a = c0.gen_scalars(n, 1, 1); b = c0.gen_scalars(n, 2, 0)
h = c0.alloc(32 * N)
for _ in range(2):
    c0.compute_h_dev(log_n, a.ptr, b.ptr, None, n, h.ptr)
ms_one = min((c0.compute_h_dev(log_n, a.ptr, b.ptr, None, n, h.ptr), c0.stats()["compute_h_ms"])[1] for _ in range(5))
ap, bp, hp, keep = [], [], [], []
for r in range(world):
    lo = min(r * M, n)
    ap.append(a.ptr + 32 * lo); bp.append(b.ptr + 32 * lo)
    d = g.ctx(r).alloc(32 * M); keep.append(d); hp.append(d.ptr)
for _ in range(2):
    g.compute_h_sharded_dev(log_n, ap, bp, None, n, hp)
best = 1e9
for _ in range(5):
    c0.sync(); t0 = time.perf_counter()
    g.compute_h_sharded_dev(log_n, ap, bp, None, n, hp)
    best = min(best, (time.perf_counter() - t0) * 1e3)
import numpy as np
want = h.download((N, 4))
got = np.concatenate([d.download((M, 4)) for d in keep])
print(f"N = 2^{log_n}, {world} ranks on ONE device: computeH on one context {ms_one:.2f} ms; over the ranks (all of them on this device, copies for all-to-alls) {best:.2f} ms "
      f"= {best / ms_one:.2f}x; h equal: {bool(np.array_equal(want, got))}")
g.close()
