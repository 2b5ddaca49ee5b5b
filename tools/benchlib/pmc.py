"""Hardware counters measured by the run itself: child `rocprofv3 --pmc` passes (bench.py, tests/test_bench_launch.py)."""
import os
import sys
import time

from .common import ROOT


def _kernel_short(name):
    """'void k_msm_accum_affine29<4, 3>(Affine<...> const*, ...)' -> 'k_msm_accum_affine29' (signature dropped; template arguments of the level-1 kernels too)"""
    k = name.split("(")[0].replace("void ", "")
    return k.split("<")[0] if k.startswith(("k_msm_accum_affine29", "k_msm_accum_affine_g2_29")) else k


def live_pmc(script, script_args, counters, timeout_s=240):
    """Hardware counters MEASURED BY THIS RUN: one child `rocprofv3 --pmc <counter>` per counter (separate passes, as
    /opt/skills/guides/MI355X_MICROARCH.md prescribes; the program itself right after `--`) over tools/<script>.  Returns
    {counter: {kernel: {"launches", "total", "per_launch"}}, "seconds": s} (values as rocprofv3 reports them, summed over the counter's dimensions:
    KB for FETCH_SIZE / WRITE_SIZE) or {"error": ...}.  Children of a process that holds the GPU are started, never exec'ed into; the parent
    is idle meanwhile (called after the timed regions)."""
    import glob
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    from collections import defaultdict
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {"error": "rocprofv3 not found"}
    t0 = time.time()
    out = {}
    tmp = tempfile.mkdtemp(prefix="live_pmc_", dir="/tmp")
    try:
        for counter in counters:
            d = os.path.join(tmp, counter)
            cmd = [exe, "--pmc", counter, "-d", d, "-o", "p", "--", sys.executable, os.path.join(ROOT, "tools", script)] + [str(x) for x in script_args]
            # (a session of its own: on a timeout the whole group goes -- the profiler AND the program under it -- not just the direct child)
            pr = subprocess.Popen(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, start_new_session=True)
            try:
                out_b, _ = pr.communicate(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                import signal
                try:
                    os.killpg(pr.pid, signal.SIGKILL)
                except OSError:
                    pass
                pr.wait()
                return {"error": f"{counter} pass over {script}: no answer within {timeout_s} s"}
            if pr.returncode != 0:
                return {"error": f"{counter} pass over {script}: rc {pr.returncode}: {out_b.decode(errors='replace')[-200:]}"}
            dbs = glob.glob(d + "/**/*_results.db", recursive=True)
            if not dbs:
                return {"error": f"{counter} pass over {script} wrote no rocpd database"}
            agg = defaultdict(lambda: [set(), 0.0])
            for k, did, v in sqlite3.connect(dbs[0]).execute("select kernel_name, dispatch_id, value from counters_collection where counter_name=?", (counter,)):
                k = _kernel_short(k); agg[k][0].add(did); agg[k][1] += v
            out[counter] = {k: {"launches": len(v[0]), "total": v[1], "per_launch": v[1] / len(v[0])} for k, v in agg.items()}
        out["seconds"] = time.time() - t0
        return out
    except Exception as e:   # a timeout, a refused profiler, an unreadable database: the line falls back to the committed passes and says so
        return {"error": f"{type(e).__name__}: {e}"[:300]}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
