"""The G2 level-1 kernel alone: a B2-shaped fixed-base G2 MSM (n pairs, c-bit windows, tables in the R' form through the public entry points)
with uniform / census / BASELINE-mix scalars.  Under `rocprofv3 --kernel-trace --stats` the level-1 launch's duration; plain, the whole MSM's.
    python3 tools/probes/solo_g2_msm.py [log_n_pairs] [c] [dist: uniform | census | whir] [reps] [knobs]"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

B = bench._binding()
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 22
cw = int(sys.argv[2]) if len(sys.argv) > 2 else 17
dist = sys.argv[3] if len(sys.argv) > 3 else "uniform"
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
n = 1 << log_n
c = B.Context(0)
for part in [x for x in (sys.argv[5] if len(sys.argv) > 5 else "").split(",") if x]:
    k, _, v = part.partition("=")
    c.set_knob(k.strip(), int(v))
nwin = (256 + cw - 1) // cw
base = c.gen_g2(n, 0x57484952 + 5)
tab = c.msm_precompute(base.ptr, n, cw, g2=True)
c.msm_table_to_rprime(tab.ptr, nwin * n, g2=True)
sc = c.gen_scalars(n, 0x57484952 + 22, bench.dist_id_of(B, dist))
c.msm_fixed_dev(tab.ptr, sc.ptr, n, cw, flags=2, g2=True)
for _ in range(reps):
    c.sync(); t0 = time.perf_counter()
    c.msm_fixed_dev(tab.ptr, sc.ptr, n, cw, flags=2, g2=True)
    dt = (time.perf_counter() - t0) * 1e3
    print(f"G2 MSM 2^{log_n} pairs, c = {cw}, {dist}: whole MSM {dt:.3f} ms", flush=True)
c.close()
