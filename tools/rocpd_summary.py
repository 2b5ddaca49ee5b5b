"""Summaries of rocprofv3's SQLite output (ROCm 7.2 writes <name>_results.db):
  python tools/rocpd_summary.py stats <results.db> <out.csv>     per-kernel calls / total / average / min / max duration (ns), like --stats
  python tools/rocpd_summary.py pmc <results.db> <out.csv>       per-kernel, per-counter sum and per-launch average (summed over the dimensions)
  python tools/rocpd_summary.py timeline <results.db> <out.txt> [window_ms]   the dispatches of the last window_ms (default 40) of the trace in start
                                                                 order: start offset, duration, queue / stream, grid, name -- one proof's critical path"""
import csv
import sqlite3
import sys
from collections import defaultdict

mode, db, out = sys.argv[1:4]
c = sqlite3.connect(db)
if mode == "timeline":
    win = float(sys.argv[4]) * 1e6 if len(sys.argv) > 4 else 40e6
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    pick = [x for x in ("start", "end", "queue_id", "stream_id", "grid_x", "grid_size_x", "workgroup_x", "workgroup_size_x", "name") if x in cols]
    rows = c.execute(f"select {', '.join(pick)} from kernels order by start").fetchall()
    t_end = max(r[1] for r in rows)
    with open(out, "w") as f:
        f.write("columns of the kernels view: " + " ".join(cols) + "\n")
        f.write("  ".join(pick) + "\n")
        for r in rows:
            if r[0] < t_end - win:
                continue
            d = dict(zip(pick, r))
            f.write(f"{(d['start'] - (t_end - win)) / 1e3:10.1f} us  {(d['end'] - d['start']) / 1e3:9.1f} us  " +
                    " ".join(str(d[k]) for k in pick[2:-1]) + "  " + d["name"].split("(")[0][:60] + "\n")
    sys.exit(0)
if mode == "stats":
    rows = c.execute("select name, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for n, k, t, a, mn, mx in rows:
            w.writerow([n, k, t, f"{a:.1f}", f"{100.0 * t / tot:.2f}", mn, mx])
    for r in rows[:14]:
        print(f"{r[0][:70]:70s} calls {r[1]:5d} total {r[2] / 1e6:9.3f} ms avg {r[3] / 1e3:9.1f} us")
else:
    cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
    agg = defaultdict(lambda: defaultdict(float)); launches = defaultdict(set)
    kn, cn, cv, did = cols.index("kernel_name"), cols.index("counter_name"), cols.index("value"), cols.index("dispatch_id")
    for r in c.execute("select * from counters_collection"):
        agg[r[kn]][r[cn]] += r[cv]; launches[r[kn]].add(r[did])
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel", "Launches", "Counter", "Sum", "PerLaunch"])
        for k, d in agg.items():
            for name, v in sorted(d.items()):
                w.writerow([k, len(launches[k]), name, v, v / len(launches[k])])
    for k, d in agg.items():
        print(k[:70], len(launches[k]), {n: round(v / len(launches[k])) for n, v in sorted(d.items())})
