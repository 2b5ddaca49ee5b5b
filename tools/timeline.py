"""Where the GPU's time goes while proofs overlap: reads the rocpd database of
`rocprofv3 --kernel-trace -d DIR -o t -- python3 bench.py --no-cpu-baseline --no-hbm-resident --sharded-msm-log-n 0 --sharded-prove-log-n 0 --steps 30`
and sweeps the kernels' [start, end) intervals of the steady middle of the run (between the 25th and 75th percentile of the level-1 launches):
for every instant, which CLASSES of kernel are resident -- issue-bound (level-1 accumulates, NTT passes), memory-bound (sorts, count / scatter),
tails (upper levels, reduces, window sums, assembly) -- and prints the share of wall time per combination.
usage: python tools/timeline.py DIR/t_results.db"""
import sqlite3
import sys

HEAVY = ("k_msm_accum_affine29", "k_msm_accum_affine_g2", "k_ntt_")   # the level-1 accumulates (G1, G2) and the NTT passes
SORT = ("k_msm2_", "k_msm_digits", "k_msm_hist", "k_msm_scatter", "k_msm_colsum", "k_scan_", "k_gather_fr", "k_max_u32")


def cls(name):
    short = name.split("(")[0].replace("void ", "")
    if short.startswith(HEAVY):
        return "issue"
    if short.startswith(SORT):
        return "memory"
    return "tail"


def main():
    c = sqlite3.connect(sys.argv[1])
    rows = c.execute("select name, start, end from kernels order by start").fetchall()
    l1 = [s for n, s, e in rows if "k_msm_accum_affine29" in n]
    lo, hi = l1[len(l1) // 4], l1[3 * len(l1) // 4]
    ev = []
    per_kernel = {}
    for n, s, e in rows:
        if e <= lo or s >= hi:
            continue
        s, e = max(s, lo), min(e, hi)
        k = cls(n)
        ev.append((s, 1, k)); ev.append((e, -1, k))
        short = n.split("(")[0].replace("void ", "")
        per_kernel[short] = per_kernel.get(short, 0) + (e - s)
    ev.sort()
    live = {"issue": 0, "memory": 0, "tail": 0}
    share = {}
    depth_time = {}
    t_prev = lo
    for t, d, k in ev:
        if t > t_prev:
            key = "+".join(x for x in ("issue", "memory", "tail") if live[x]) or "idle"
            share[key] = share.get(key, 0) + (t - t_prev)
            nlive = sum(live.values())
            depth_time[nlive] = depth_time.get(nlive, 0) + (t - t_prev)
            t_prev = t
        live[k] += d
    wall = hi - lo
    print(f"window {wall / 1e6:.1f} ms, {len(ev) // 2} launches")
    for k, v in sorted(share.items(), key=lambda kv: -kv[1]):
        print(f"  {k:22s} {100.0 * v / wall:6.2f} %")
    print("kernels resident at once:", "  ".join(f"{n}: {100.0 * v / wall:.1f} %" for n, v in sorted(depth_time.items())))
    print("busiest kernels (sum of durations / wall):")
    for k, v in sorted(per_kernel.items(), key=lambda kv: -kv[1])[:14]:
        print(f"  {k[:60]:60s} {v / wall:6.3f} {cls(k)}")


if __name__ == "__main__":
    main()
