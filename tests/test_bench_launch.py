"""bench.py --gpus N launched PLAINLY (no torch.distributed.run) must start its N rank processes itself -- children of a process
that never touches the GPU -- and relay rank 0's one JSON line (the driver's multi-GPU command may have exactly the
shape of its 1-GPU command).  --launch-check stops every rank after the rendezvous (gloo, no GPU), so this runs on CPU."""
import json
import os
import subprocess
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("n", [2, 3, 8])
def test_plain_launch_starts_its_ranks_and_prints_one_line(n):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--launch-check"], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout
    j = json.loads(lines[0])
    assert j == {"launch_check": True, "world": n, "sum_of_ranks": float(n * (n - 1) // 2)}


def test_more_ranks_than_distinct_gpus_is_refused_with_a_message_unless_it_is_the_declared_rehearsal():
    """`--gpus N` means N ranks on N distinct devices.  A node with fewer GPUs (this container has none; a one-GPU box has
    one) refuses a plain `--gpus 2` with that message and a non-zero status BEFORE any rank starts; the one-GPU rehearsal has to be asked for
    by name (and then fails later here, for want of any GPU -- with the 'no CPU path' message, not the device-count one)."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs present")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    bench = os.path.join(ROOT, "bench.py")
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and "distinct_devices != n_gpus" in r.stderr and "--rehearse-on-one-gpu" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")], "no JSON line may come out of a refused run"
    if torch.cuda.device_count() == 0:
        r = subprocess.run([sys.executable, bench, "--gpus", "2", "--rehearse-on-one-gpu", "--steps", "1", "--warmup", "0", "--sharded-msm-log-n", "0", "--sharded-prove-log-n", "0"],
                           capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode != 0 and "distinct_devices" not in r.stderr


def test_launcher_given_ranks_are_used_as_they_are():
    """under torch.distributed.run (WORLD_SIZE set) bench.py must NOT start ranks of its own"""
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29517")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--launch-check"], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    assert json.loads(out.stdout.strip().splitlines()[-1])["world"] == 1
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"], capture_output=True, text=True, timeout=300, env=env)
    assert bad.returncode != 0 and "WORLD_SIZE=1" in bad.stderr


def _bench_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_for_tests", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    return m


def test_kernel_names_of_the_counter_passes_are_folded_like_the_committed_summaries():
    b = _bench_module()
    assert b._kernel_short("void k_msm_accum_affine29<4, 3>(Affine<Fe<FpParams> > const*, unsigned int const*)") == "k_msm_accum_affine29"
    assert b._kernel_short("void k_msm_accum_affine_g2_29<1>(Affine<Fp2> const*)") == "k_msm_accum_affine_g2_29"
    assert b._kernel_short("void k_msm2_partition<20u>(Msm2Shape, Fe<FrParams> const*)") == "k_msm2_partition<20u>"
    assert b._kernel_short("k_ntt_pass_wave(Fe<FrParams>*, Fe<FrParams> const*, NttPass, NttTables)") == "k_ntt_pass_wave"


@pytest.mark.gpu
def test_live_counter_pass_sees_the_level1_launch():
    """bench.py's `roofline.traffic` is measured by the run itself: a child `rocprofv3 --pmc FETCH_SIZE` over tools/solo_z_msm.py (small here)"""
    if "rocprof" in os.environ.get("LD_PRELOAD", "") or any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ):
        pytest.skip("this process already runs under a profiler: profilers are not nested")
    b = _bench_module()
    r = b.live_pmc("solo_z_msm.py", [16, 1], ("FETCH_SIZE",), timeout_s=300)
    assert "error" not in r, r
    k = r["FETCH_SIZE"]["k_msm_accum_affine29"]
    # the fixed-base launch gathers one 64-B point per addition: at least the scalars' 32 B per pair reach the kernel (KB units)
    assert k["launches"] == 2 and k["per_launch"] * 1024.0 > 32.0 * ((1 << 16) - 1)


def test_clock_sampler_reads_rocm_smi_csv(monkeypatch):
    """bench.py's ClockSampler: one sample per rocm-smi answer (csv header + one row), nothing on garbage or a missing tool"""
    b = _bench_module()
    import subprocess as sp
    good = ("device,fclk clock speed:,fclk clock level:,mclk clock speed:,mclk clock level:,sclk clock speed:,sclk clock level:,socclk clock speed:,socclk clock level:,"
            "Current Socket Graphics Package Power (W)\ncard0,(1250Mhz),0,(2000Mhz),0,(2177Mhz),1,(1200Mhz),0,1284.0\n")
    answers = [good, "", "no such tool", good.replace("2177", "2201").replace("1284.0", "1300.0")]

    class R:
        def __init__(self, out): self.stdout = out

    def fake_run(cmd, **kw):
        assert cmd[:3] == ["rocm-smi", "-d", "0"]
        if not answers:
            raise FileNotFoundError("rocm-smi")
        return R(answers.pop(0))
    monkeypatch.setattr(sp, "run", fake_run)
    s = b.ClockSampler(0)
    for _ in range(5):
        s._once()
    assert s.samples == [(2177, 1284.0), (2201, 1300.0)]
    s._stop.set(); s._th.start()
    r = s.stop()
    assert r["samples"] == 2 and r["sclk_mhz_min"] == 2177 and r["sclk_mhz_max"] == 2201 and abs(r["package_power_w_mean"] - 1292.0) < 1e-9
