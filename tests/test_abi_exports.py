"""CPU suite: the C-ABI shared library loads and exports every symbol include/mi355x_groth16.h
declares (no compute calls: there is no GPU here), and its host-only entry points (encoding,
partial-sum combine) agree with the oracle."""
import ctypes as C
import os
import re
import numpy as np
import pytest
import pyref as P
import cref
from helpers import *
from gpu_common import load_binding, ROOT


HEADERS = ("mi355x_groth16.h", "mi355x_groth16_group.h", "mi355x_whir_ingest.h", "mi355x_groth16_debug.h")


def _declared(headers=HEADERS):
    out = set()
    for h in headers:
        src = open(os.path.join(ROOT, "include", h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)   # (comments mention entry points of the other headers)
        out |= set(re.findall(r"\b(mi_[a-z0-9_]+)\s*\(", src))
    return sorted(out)


def test_product_header_carries_no_lab_bench():
    """what a maintainer binds (include/mi355x_groth16.h, + _group.h / whir_ingest.h by need) declares no generator, probe,
    plan knob or fault injection -- those live in mi355x_groth16_debug.h, which the Go shim never includes -- and stays short"""
    for h in HEADERS[:3]:
        names = _declared((h,))
        bad = [n for n in names if n.startswith(("mi_debug_", "mi_bench_", "mi_gen_")) or n in ("mi_field_op_dev", "mi_g1_add_dev", "mi_g2_add_dev")]
        assert not bad, f"{h} declares lab-bench symbols: {bad}"
    assert len(open(os.path.join(ROOT, "include", HEADERS[0])).read().splitlines()) < 350
    dbg = _declared((HEADERS[3],))
    assert "mi_debug_inject_hip_failure" in dbg and "mi_gen_scalars_dev" in dbg and "mi_bench_modmul_dev" in dbg
    go = os.path.join(ROOT, "gnark-whir_amd", "go")
    for d, _, fs in os.walk(go):
        for f in fs:
            if f.endswith(".go"):
                assert "mi355x_groth16_debug.h" not in open(os.path.join(d, f)).read(), f"{f} includes the debug header"


def test_library_exports_every_declared_symbol():
    B = load_binding()
    lib = B.load()
    names = _declared()
    assert len(names) >= 35
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/*.h but not exported"
    for n in B.EXPORTS:
        assert n in names


def test_no_device_means_hard_failure_not_fallback():
    """without a gfx950 device mi_init must fail (MI_ENODEV); the binding raises: there is no CPU path"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    B = load_binding()
    with pytest.raises(B.MiError):
        B.Context(0)
    with pytest.raises(B.MiError):      # the prover pool is a set of contexts: same hard failure, nothing leaked, no thread left behind
        B.Prover(0, 3)
    import ctypes as C
    h = C.c_void_p()
    assert B.load().mi_init_prio(0, 7, C.byref(h)) != 0 and not h.value          # unknown priority scheme
    assert B.load().mi_prover_create(0, 0, C.byref(h)) != 0 and not h.value       # in_flight out of range


def test_host_encoding_matches_oracle():
    B = load_binding()
    g1 = cref.gen_g1(20, 3); g2 = cref.gen_g2(6, 4)
    for row in list(g1) + [np.zeros(8, np.uint64)]:
        assert B.g1_compress(row) == cref.g1_compress(row)
    for row in list(g2) + [np.zeros(16, np.uint64)]:
        assert B.g2_compress(row) == cref.g2_compress(row)
    assert B.g1_compress(g1_arr([P.G1_GEN])[0]).hex() == "80" + "00" * 30 + "01"
    raw = np.concatenate([g1[0], g2[0], g1[1]])
    assert B.proof_write(raw) == cref.proof_write(raw)
    assert B.proof_write(raw, commitments=g1[2:4].copy(), pok=g1[5].copy()) == cref.proof_write(raw, commitments=g1[2:4].copy(), pok=g1[5].copy())
    assert len(B.proof_write(raw, commitments=g1[2:3].copy(), pok=g1[5].copy())) == 196   # the WHIR circuit's proof size


def test_host_partial_sum_combine_matches_oracle():
    B = load_binding()
    pts = cref.gen_g1(9, 8); sc = cref.gen_scalars(9, 9, 0)
    parts = np.stack([cref.msm_g1(pts[i:i + 3], sc[i:i + 3]) for i in (0, 3, 6)])
    assert np.array_equal(B.g1_sum(parts), cref.msm_g1(pts, sc))
    inf = np.zeros(12, np.uint64); inf[:8] = parts[0][:8]   # Z = 0 -> infinity whatever X, Y hold
    assert np.array_equal(B.g1_sum(np.stack([parts[0], inf])), parts[0])
    p2 = cref.gen_g2(4, 8); s2 = cref.gen_scalars(4, 9, 0)
    parts2 = np.stack([cref.msm_g2(p2[:2], s2[:2]), cref.msm_g2(p2[2:], s2[2:])])
    assert np.array_equal(B.g2_sum(parts2), cref.msm_g2(p2, s2))


def test_host_pedersen_fold_matches_oracle():
    B = load_binding()
    pts = cref.gen_g1(4, 81); ch = cref.gen_scalars(1, 82, 0)[0]
    assert np.array_equal(B.pedersen_fold(pts, ch), cref.pedersen_fold(pts, ch))
    assert np.array_equal(B.pedersen_fold(pts[:1], ch), pts[0])          # challenge^0 = 1


def test_bench_refuses_to_run_without_a_gpu_and_its_helper_idles_quietly():
    """bench.py has no CPU path (it must say so, not fall back), and its sharded-MSM helper process -- started before the parent touches
    the GPU -- ends without a word when the parent never sends it work"""
    import subprocess, sys
    bench = os.path.join(ROOT, "bench.py")
    r = subprocess.run([sys.executable, bench, "--sharded-helper"], stdin=subprocess.DEVNULL, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip() == ""
    r = subprocess.run([sys.executable, bench, "--steps", "1", "--warmup", "0", "--sharded-msm-log-n", "0"], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "no CPU path" in (r.stderr + r.stdout)
