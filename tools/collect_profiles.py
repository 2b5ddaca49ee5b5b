"""Turns the raw outputs of the round's final measurement run (gpurun_out/r3_*) into the committed summaries under profiles/.
usage: python tools/collect_profiles.py"""
import json
import sqlite3
import subprocess
import sys
from collections import defaultdict

G = "gpurun_out/"


def line(path):
    return [x for x in open(path) if x.startswith("{")][-1]


for src, dst in ((G + "r3_bench_final.log", "profiles/r03_bench_line_final.json"), (G + "r3_prof_def.log", "profiles/r03_bench_line_profiled_default.json"),
                 (G + "r3_prof_if1.log", "profiles/r03_bench_line_profiled_inflight1.json"), (G + "r3_bench26_final.log", "profiles/r03_bench_line_N2p26.json")):
    l = line(src); open(dst, "w").write(l); d = json.loads(l)
    print(dst, {k: round(d[k], 3) if isinstance(d[k], float) else d[k] for k in ("value", "ms_per_step", "single_proof_latency_ms", "value_host_inputs", "hbm_in_use_gb")},
          "launch_ms", round(d["roofline"]["launch_ms"], 2), "frac", round(d["roofline"]["frac"], 4), "ntt frac", round(d["roofline_ntt"]["frac"], 4))
subprocess.check_call([sys.executable, "tools/rocpd_summary.py", "stats", G + "r3_prof_def/d_results.db", "profiles/r03_kernel_stats_default.csv"], stdout=subprocess.DEVNULL)
subprocess.check_call([sys.executable, "tools/rocpd_summary.py", "stats", G + "r3_prof_if1/i_results.db", "profiles/r03_kernel_stats_inflight1.csv"], stdout=subprocess.DEVNULL)
out = {}
for name, db in (("FETCH_SIZE", G + "r3_pmc_fetch/f_results.db"), ("WRITE_SIZE", G + "r3_pmc_write/w_results.db")):
    c = sqlite3.connect(db)
    agg = defaultdict(lambda: [set(), 0.0])
    for k, did, v in c.execute("select kernel_name, dispatch_id, value from counters_collection where counter_name=?", (name,)):
        k = k.split("(")[0]; agg[k][0].add(did); agg[k][1] += v
    out[name] = {k: {"launches": len(v[0]), "kb_total": v[1], "kb_per_launch": v[1] / len(v[0])} for k, v in agg.items()}
out["_note"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over `python3 bench.py --in-flight 1 --steps 2 --warmup 1 --no-cpu-baseline "
                "--no-host-inputs --sharded-msm-log-n 0 --sharded-prove-log-n 0`, N=2^23, final round-3 code; KB as rocprofv3 reports them, summed over the counter's dimensions; FETCH_SIZE raw "
                "(k_bench_gather: 33.5 M random 64-B gathers read 2.10 GB raw, so 64-B gathers need no correction; 16-B-per-lane streams need x2)")
json.dump(out, open("profiles/r03_pmc_bench_traffic.json", "w"), indent=1)
for k in ("k_msm_accum_affine29", "k_ntt_pass_wave", "k_bench_gather"):
    print(k, "fetch MB/launch", round(out["FETCH_SIZE"][k]["kb_per_launch"] / 1e3, 1), "write", round(out["WRITE_SIZE"].get(k, {"kb_per_launch": 0})["kb_per_launch"] / 1e3, 1))
c = sqlite3.connect(G + "r3_prof_if1/i_results.db")
for n, k, a in c.execute("select name, count(*), avg(end-start) from kernels where name like 'k_msm_accum_affine29%' or name like 'k_ntt_pass_wave%' group by name"):
    print("in-flight 1 rocprof average:", n[:30], k, round(a / 1e6, 3), "ms")
