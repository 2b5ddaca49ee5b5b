"""Spawns groups of rank processes for the multi-process GPU tests (tests/test_gpu_group_multiprocess.py).

Started by tests/conftest.py at session start -- BEFORE the pytest process makes its first GPU call -- and never touches the GPU
itself, so it can fork and exec at any time (a process that has initialised the GPU must not exec; pytest's own process has, by the
time the multi-process tests run).  Protocol: one JSON request per line on stdin
    {"argv": [...], "world": N, "env": {...}, "timeout": seconds}
-> N child processes with MI_RANK / MI_WORLD set, stdout / stderr into temporary files -> one JSON answer per line on stdout
    {"ranks": [{"rc": int | "timeout", "seconds": float, "stdout": "...", "stderr": "... (tail)"}]}
Test infrastructure only."""
import json
import os
import subprocess
import sys
import tempfile
import time


def run(req):
    world = int(req["world"])
    timeout = float(req.get("timeout", 300))
    procs = []
    t0 = time.time()
    for r in range(world):
        env = dict(os.environ)
        env.update(req.get("env", {}))
        env.update({"MI_RANK": str(r), "MI_WORLD": str(world)})
        fo, fe = tempfile.TemporaryFile("w+"), tempfile.TemporaryFile("w+")
        procs.append((subprocess.Popen(req["argv"], env=env, stdout=fo, stderr=fe, stdin=subprocess.DEVNULL), fo, fe))
    out = []
    for p, fo, fe in procs:
        left = max(0.1, timeout - (time.time() - t0))
        try:
            rc = p.wait(timeout=left)
        except subprocess.TimeoutExpired:
            p.kill()
            p.wait()
            rc = "timeout"
        fo.seek(0)
        fe.seek(0)
        out.append({"rc": rc, "seconds": time.time() - t0, "stdout": fo.read()[-20000:], "stderr": fe.read()[-4000:]})
        fo.close()
        fe.close()
    return {"ranks": out}


def main():
    for line in sys.stdin:
        line = line.strip()
        if not line:
            continue
        try:
            ans = run(json.loads(line))
        except BaseException as e:   # the answer must come, whatever happened
            ans = {"error": f"{type(e).__name__}: {e}"}
        print(json.dumps(ans), flush=True)


if __name__ == "__main__":
    main()
