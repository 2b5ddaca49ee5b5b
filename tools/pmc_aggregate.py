"""Aggregate rocprofv3 --pmc counter CSVs (one pass per counter) into per-kernel totals.
usage: python tools/pmc_aggregate.py out.json FETCH_SIZE=<counter_collection.csv> WRITE_SIZE=<counter_collection.csv>
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KB (MI355X_MICROARCH.md, HBM section); no correction applied here."""
import csv, json, sys
from collections import defaultdict

out = {}
for arg in sys.argv[2:]:
    name, path = arg.split("=", 1)
    agg = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") != name:
            continue
        k = r["Kernel_Name"].split("(")[0]
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
    out[name] = {k: {"launches": v[0], "kb_total": v[1], "kb_per_launch": v[1] / v[0]} for k, v in agg.items()}
json.dump(out, open(sys.argv[1], "w"), indent=1)
for name, d in out.items():
    for k, v in sorted(d.items(), key=lambda kv: -kv[1]["kb_total"])[:8]:
        print(name, k[:60], v["launches"], round(v["kb_per_launch"] / 1e3, 1), "MB/launch")
