"""The launch bench.py's `roofline` line is quoted on, alone: a Z-shaped fixed-base G1 MSM (N - 1 uniform scalars, c = 20, tables in the R'
form through the public entry points).  Under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes) its level-1 launch's HBM
traffic; plain, its duration.   python3 tools/solo_z_msm.py [log_n] [reps] [knobs: name=value,...]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

B = bench._binding()
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 23
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
N = 1 << log_n
c = B.Context(0)
for part in [x for x in (sys.argv[3] if len(sys.argv) > 3 else "").split(",") if x]:
    k, _, v = part.partition("=")
    c.set_knob(k.strip(), int(v))
cz, n = 20, N - 1
nwin = (256 + cz - 1) // cz
base = c.gen_g1(N, 0x57484952 + 5)
tab = c.msm_precompute(base.ptr, n, cz)
c.msm_table_to_rprime(tab.ptr, nwin * n)
sc = c.gen_scalars(n, 0x57484952 + 22, 0)
c.msm_fixed_dev(tab.ptr, sc.ptr, n, cz, flags=2)
for _ in range(reps):
    c.msm_fixed_dev(tab.ptr, sc.ptr, n, cz, flags=2)
    st = c.stats()
    print(f"level-1 launch {st['g1_accum_kernel_ms']:.3f} ms, {st['g1_accum_entries']} additions = {st['g1_accum_entries'] / st['g1_accum_kernel_ms'] / 1e6:.2f} G/s, "
          f"algorithmic {96.0 * n / st['g1_accum_kernel_ms'] / 1e6:.1f} GB/s, whole MSM {st['total_ms']:.3f} ms", flush=True)
c.close()
