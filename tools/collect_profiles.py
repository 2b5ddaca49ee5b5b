"""Turns the raw outputs of the round's final measurement run (gpurun_out/r6_*, tools/final_measure.sh) into the committed summaries under
profiles/.  usage: python tools/collect_profiles.py"""
import glob
import hashlib
import json
import os
import sqlite3
import subprocess
import sys
from collections import defaultdict

G = "gpurun_out/"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def line(path):
    return [x for x in open(path) if x.startswith("{")][-1]


def db(d):
    return glob.glob(G + d + "/**/*_results.db", recursive=True)[0]


def short(kernel_name):
    """'void k_msm_accum_affine29<4, 3>(Affine<...> const*, ...)' -> 'k_msm_accum_affine29' (template arguments and the signature dropped)"""
    k = kernel_name.split("(")[0].replace("void ", "")
    return k.split("<")[0] if k.startswith(("k_msm_accum_affine29", "k_msm_accum_affine_g2_29")) else k


def sha16(path):
    return hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]


for src, dst in ((G + "r6_bench_final.log", "profiles/r06_bench_line_final.json"), (G + "r6_prof_def.log", "profiles/r06_bench_line_profiled_default.json"),
                 (G + "r6_prof_if1.log", "profiles/r06_bench_line_profiled_inflight1.json"), (G + "r6_bench26_final.log", "profiles/r06_bench_line_N2p26.json"),
                 (G + "r6_rehearse2.log", "profiles/r06_bench_line_rehearsal_2ranks_one_gpu.json")):
    if not os.path.exists(src):
        print("missing", src); continue
    l = line(src); open(dst, "w").write(l); d = json.loads(l)
    print(dst, {k: round(d[k], 3) if isinstance(d.get(k), float) else d.get(k) for k in ("value", "ms_per_step", "value_hbm_resident_inputs", "single_proof_latency_ms", "single_proof_latency_host_inputs_ms", "hbm_in_use_gb")},
          "launch_ms", round(d["roofline"]["launch_ms"], 2), "frac", round(d["roofline"]["frac"], 4), "ntt frac", round(d["roofline_ntt"]["frac"], 4))
for d_, out in (("r6_prof_def", "profiles/r06_kernel_stats_default.csv"), ("r6_prof_if1", "profiles/r06_kernel_stats_inflight1.csv"),
                ("r6_prof_solo_z", "profiles/r06_kernel_stats_solo_z.csv")):   # (solo_z: the launch the roofline line is quoted on, alone: its average must agree with roofline.launch_ms)
    if os.path.isdir(G + d_):
        subprocess.check_call([sys.executable, "tools/rocpd_summary.py", "stats", db(d_), out], stdout=subprocess.DEVNULL)
if os.path.isdir(G + "r6_pmc_fetch") and os.path.isdir(G + "r6_pmc_write"):
    out = {}
    for name, d_ in (("FETCH_SIZE", "r6_pmc_fetch"), ("WRITE_SIZE", "r6_pmc_write")):
        c = sqlite3.connect(db(d_))
        agg = defaultdict(lambda: [set(), 0.0])
        for k, did, v in c.execute("select kernel_name, dispatch_id, value from counters_collection where counter_name=?", (name,)):
            k = short(k); agg[k][0].add(did); agg[k][1] += v
        out[name] = {k: {"launches": len(v[0]), "kb_total": v[1], "kb_per_launch": v[1] / len(v[0])} for k, v in agg.items()}
    csrc = os.path.join(ROOT, "gnark-whir_amd", "csrc")
    out["_sources"] = {f: sha16(os.path.join(csrc, f)) for f in ("msm.hip", "msm_g1.hip", "msm_core.cuh", "msm2_core.cuh", "curve29.cuh", "field29.cuh", "ntt.hip", "ntt_tile.cuh", "ntt_wave.cuh", "field.cuh")}
    out["_note"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over `python3 bench.py --in-flight 1 --steps 2 --warmup 1 --no-cpu-baseline --no-hbm-resident "
                    "--sharded-msm-log-n 0 --sharded-prove-log-n 0 --no-sensitivity --no-solo-legs --n-committed 0` (no commitment: the per-launch averages are those of the proof's own MSMs), N=2^23, final round-5 code (_sources: sha256[:16] of the kernel sources the passes ran on; bench.py withholds "
                    "`traffic` when they differ); KB as rocprofv3 reports them, summed over the counter's dimensions; FETCH_SIZE raw (64-B gathers need no correction; 16-B-per-lane streams need x2)")
    if os.path.isdir(G + "r6_pmc_solo_fetch") and os.path.isdir(G + "r6_pmc_solo_write"):   # the solo Z-shaped launch (tools/solo_z_msm.py): the roofline line's basis
        solo = {}
        for name, d_ in (("FETCH_SIZE", "r6_pmc_solo_fetch"), ("WRITE_SIZE", "r6_pmc_solo_write")):
            c = sqlite3.connect(db(d_))
            tot, ids = 0.0, set()
            for k, did, v in c.execute("select kernel_name, dispatch_id, value from counters_collection where counter_name=?", (name,)):
                if short(k) == "k_msm_accum_affine29":
                    tot += v; ids.add(did)
            solo[name + "_kb"] = tot / max(1, len(ids)); solo["launches_" + name] = len(ids)
        solo["what"] = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over `python3 tools/solo_z_msm.py 23 2`: KB per launch of k_msm_accum_affine29 (the Z-shaped fixed-base MSM alone on the GPU)"
        out["solo_z"] = solo
    json.dump(out, open("profiles/r06_pmc_bench_traffic.json", "w"), indent=1)
    for k in ("k_msm_accum_affine29", "k_ntt_pass_wave", "k_msm2_scatter2_staged"):
        if k not in out["FETCH_SIZE"]:
            continue
        print(k, "fetch MB/launch", round(out["FETCH_SIZE"][k]["kb_per_launch"] / 1e3, 1), "write", round(out["WRITE_SIZE"].get(k, {"kb_per_launch": 0})["kb_per_launch"] / 1e3, 1))
if os.path.isdir(G + "r6_pmc_valu"):
    subprocess.check_call([sys.executable, "tools/rocpd_summary.py", "pmc", db("r6_pmc_valu"), "profiles/r06_pmc_valu_proofs.csv"], stdout=subprocess.DEVNULL)
p = G + "r6_bench_final.log"
if os.path.exists(p):
    d = json.loads(line(p))
    sens = d.get("sensitivity")
    if sens:
        g1 = d["g1_msm_solo"]
        rows = [f"witness `whir`         (45 % {{0,1}} / 25 % bytes / 5 % 64-bit / 25 % uniform): value {d['value']:.2f} proofs/s, HBM-resident {d['value_hbm_resident_inputs']:.2f}, "
                f"single proof {d['single_proof_latency_ms']:.2f} ms"]
        for key in ("half_uniform", "uniform", "census"):
            x = sens[key]
            rows.append(f"witness `{key}`".ljust(22) + f": value {x['value']:.2f} proofs/s, HBM-resident {x['value_hbm_resident_inputs']:.2f}, single proof {x['single_proof_latency_ms']:.2f} ms, "
                        f"G1 level-1 additions per proof {x['g1_level1_additions_per_proof'] / 1e6:.1f} M")
        open("profiles/r06_scalar_mix.txt", "w").write("the `sensitivity` block of profiles/r06_bench_line_final.json (python bench.py, defaults: N = 2^23, three proofs in flight, the step = Commit + host-input prove "
                                                        "with the PoK; same key, same box, same run; every proof compared with its leg's reference proof)\n" + "\n".join(rows) + "\n" +
                                                        "census = " + sens["census"].get("mix", "") + "\n")
        print("\n".join(rows))

for m in ("census", "whir"):
    src = G + f"r6_prof_{m}_summary.txt"
    if os.path.exists(src):
        open(f"profiles/r06_kernel_profile_{m}.txt", "w").write(f"rocprofv3 --kernel-trace over `python3 tools/prof_proof.py 23 8 x {m}`: kernel time per proof, ONE proof at a time on one context, inputs in HBM "
                                                                 "(launch durations summed / 8; kernels of different streams overlap, so the column does not add up to the proof's latency)\n" + open(src).read())
