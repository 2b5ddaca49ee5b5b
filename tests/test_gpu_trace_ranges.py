"""mi_debug_set_trace_ranges ("roctx ranges around NTT/MSM phases"): the ranges reach rocprofv3's marker trace, every
host-side phase of a proof is there once per proof, and switching them on changes no result."""
import glob
import json
import os
import shutil
import sqlite3
import subprocess
import sys
import tempfile
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

PHASES = ["mi.prove", "mi.prove.blinding", "mi.prove.assemble", "mi.computeH.a.enqueue", "mi.computeH.b.enqueue", "mi.computeH.c.enqueue", "mi.computeH.last.enqueue"] + \
         [f"mi.msm.{m}.{what}" for m in ("A", "B1", "B2", "K", "Z") for what in ("enqueue", "collect")]


def test_ranges_reach_the_marker_trace_once_per_proof():
    if "rocprof" in os.environ.get("LD_PRELOAD", "") or any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ):
        pytest.skip("this process already runs under a profiler: profilers are not nested")
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    tmp = tempfile.mkdtemp(prefix="ranges_", dir="/tmp")
    try:
        # (the program itself right after `--`; a child of this process, which holds the GPU, never an exec)
        r = subprocess.run([exe, "--marker-trace", "--kernel-trace", "-d", tmp, "-o", "m", "--", sys.executable, os.path.join(ROOT, "tools", "prof_proof.py"), "14", "3", "ranges"],
                           cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
        assert r.returncode == 0, r.stdout.decode(errors="replace")[-1500:]
        db = glob.glob(tmp + "/**/*_results.db", recursive=True)
        assert db, "rocprofv3 wrote no rocpd database"
        c = sqlite3.connect(db[0])
        seen = {}
        for (ext,) in c.execute("select extdata from regions where category = 'MARKER_CORE_RANGE_API'"):
            m = json.loads(ext).get("message", "")
            seen[m] = seen.get(m, 0) + 1
        assert {k: seen.get(k, 0) for k in PHASES} == {k: 3 for k in PHASES}, seen
        # the proof's range encloses its phases: every phase range of a thread that also opened mi.prove starts inside one
        spans = [(s, e) for s, e, ext in c.execute("select start, end, extdata from regions where category = 'MARKER_CORE_RANGE_API'") if json.loads(ext).get("message") == "mi.prove"]
        inner = [(s, e) for s, e, ext in c.execute("select start, end, extdata from regions where category = 'MARKER_CORE_RANGE_API'") if json.loads(ext).get("message") == "mi.msm.Z.enqueue"]
        assert all(any(a <= s and e <= b for a, b in spans) for s, e in inner)
        assert c.execute("select count(*) from kernels").fetchone()[0] > 0
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def test_switch_is_idempotent_and_results_do_not_change():
    import numpy as np
    from gpu_common import load_binding
    B = load_binding()
    lib = B.load()
    c = B.Context(0)
    try:
        n = 1 << 12
        x = c.gen_scalars(n, 7, 0)
        base = c.gen_g1(n, 11)
        r_off = c.msm_g1_dev(base.ptr, x.ptr, n)
        assert lib.mi_debug_set_trace_ranges(1) == 0 and lib.mi_debug_set_trace_ranges(1) == 0
        r_on = c.msm_g1_dev(base.ptr, x.ptr, n)
        assert lib.mi_debug_set_trace_ranges(0) == 0
        assert np.array_equal(r_on, r_off) and np.array_equal(c.msm_g1_dev(base.ptr, x.ptr, n), r_off)
    finally:
        lib.mi_debug_set_trace_ranges(0)
        c.close()
