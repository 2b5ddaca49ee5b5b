import json
import os
import subprocess
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


class RankLauncher:
    """The process that spawns rank processes for the multi-process GPU tests (tests/rank_launcher.py).  It is started at session start,
    before this process makes its first GPU call (a process that has initialised the GPU must not exec, and a fork of it is the
    same process image), and stays idle until a test asks."""

    def __init__(self):
        self.p = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "rank_launcher.py")], stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True)

    def run(self, argv, world, env=None, timeout=300):
        self.p.stdin.write(json.dumps({"argv": argv, "world": world, "env": env or {}, "timeout": timeout}) + "\n")
        self.p.stdin.flush()
        line = self.p.stdout.readline()
        if not line:
            raise RuntimeError("the rank launcher ended without an answer")
        ans = json.loads(line)
        if "error" in ans:
            raise RuntimeError("rank launcher: " + ans["error"])
        return ans["ranks"]

    def close(self):
        try:
            self.p.stdin.close()
            self.p.wait(timeout=10)
        except BaseException:
            self.p.kill()


_LAUNCHER = None


def pytest_sessionstart(session):
    global _LAUNCHER
    expr = session.config.getoption("markexpr", "") or ""
    if "gpu" in expr and "not gpu" not in expr:   # a GPU session: start the launcher now, while this process is still GPU-free
        _LAUNCHER = RankLauncher()


def pytest_sessionfinish(session, exitstatus):
    global _LAUNCHER
    if _LAUNCHER is not None:
        _LAUNCHER.close()
        _LAUNCHER = None


@pytest.fixture(scope="session")
def rank_launcher():
    global _LAUNCHER
    if _LAUNCHER is None:   # (a session that was not started with -m gpu: best effort, the process may already hold the GPU)
        _LAUNCHER = RankLauncher()
    return _LAUNCHER
