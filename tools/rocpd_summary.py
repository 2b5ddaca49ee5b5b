"""Summaries of rocprofv3's SQLite output (ROCm 7.2 writes <name>_results.db):
  python tools/rocpd_summary.py stats <results.db> <out.csv>     per-kernel calls / total / average / min / max duration (ns), like --stats
  python tools/rocpd_summary.py pmc <results.db> <out.csv>       per-kernel, per-counter sum and per-launch average (summed over the dimensions)"""
import csv
import sqlite3
import sys
from collections import defaultdict

mode, db, out = sys.argv[1:4]
c = sqlite3.connect(db)
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
