"""Is the job running at the 2.4 GHz the roofline figures assume?  Samples rocm-smi (read-only: clocks, package power, temperatures) while a
child process runs (a) nothing, (b) bench.py's proofs (three in flight), (c) the solo Z-shaped level-1 launch in a loop, (d) computeH in a loop.
This process never touches the GPU.   python3 tools/probes/clocks_under_load.py"""
import os
import re
import subprocess
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def sample():
    out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp", "--csv"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True).stdout
    rows = [r for r in out.splitlines() if r.strip()]
    if len(rows) < 2:
        return None
    head, val = rows[0].split(","), rows[1].split(",")
    d = dict(zip(head, val))
    mhz = lambda k: int(re.sub(r"[^0-9]", "", d.get(k, "0")) or 0)
    return {"sclk": mhz("sclk clock speed:"), "mclk": mhz("mclk clock speed:"), "fclk": mhz("fclk clock speed:"),
            "power_w": float(d.get("Current Socket Graphics Package Power (W)", "0") or 0), "t_junction": float(d.get("Temperature (Sensor junction) (C)", "0") or 0)}


def watch(name, cmd, warm_s, n=12):
    p = subprocess.Popen(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) if cmd else None
    time.sleep(warm_s)
    xs = []
    for _ in range(n):
        if p is not None and p.poll() is not None:
            break
        s = sample()
        if s:
            xs.append(s)
        time.sleep(0.4)
    if p is not None:
        p.wait()
    if not xs:
        print(f"{name}: no samples (the workload ended before the window)"); return
    avg = lambda k: sum(x[k] for x in xs) / len(xs)
    print(f"{name:42s} {len(xs):2d} samples: sclk {min(x['sclk'] for x in xs)}-{max(x['sclk'] for x in xs)} MHz (mean {avg('sclk'):.0f}), mclk {avg('mclk'):.0f}, "
          f"package power {avg('power_w'):.0f} W, junction {avg('t_junction'):.0f} C", flush=True)


cap = subprocess.run(["rocm-smi", "--showmaxpower"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True).stdout
print(" ".join(x.strip() for x in cap.splitlines() if "Max" in x or "max" in x))
py = sys.executable
watch("idle", None, 0.5, 4)
watch("bench.py proofs (three in flight)", [py, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--sharded-msm-log-n", "0", "--sharded-prove-log-n", "0", "--no-sensitivity",
                                            "--no-live-pmc", "--no-solo-legs", "--no-hbm-resident", "--steps", "400"], 7.0)
watch("solo Z-shaped fixed-base MSM, looped", [py, os.path.join(ROOT, "tools", "solo_z_msm.py"), "23", "900"], 7.0)
watch("solo computeH, looped", [py, os.path.join(ROOT, "tools", "ntt_probe.py"), "23", "1500"], 5.0)
