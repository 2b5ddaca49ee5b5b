"""bench.py --gpus N launched PLAINLY (no torch.distributed.run) must start its N rank processes itself -- children of a process
that never touches the GPU -- and relay rank 0's one JSON line (the driver's multi-GPU command may have exactly the
shape of its 1-GPU command).  --launch-check stops every rank after the rendezvous (gloo, no GPU), so this runs on CPU."""
import json
import os
import subprocess
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("n", [2, 3])
def test_plain_launch_starts_its_ranks_and_prints_one_line(n):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--launch-check"], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout
    j = json.loads(lines[0])
    assert j == {"launch_check": True, "world": n, "sum_of_ranks": float(n * (n - 1) // 2)}


def test_launcher_given_ranks_are_used_as_they_are():
    """under torch.distributed.run (WORLD_SIZE set) bench.py must NOT start ranks of its own"""
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29517")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--launch-check"], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    assert json.loads(out.stdout.strip().splitlines()[-1])["world"] == 1
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"], capture_output=True, text=True, timeout=300, env=env)
    assert bad.returncode != 0 and "WORLD_SIZE=1" in bad.stderr
