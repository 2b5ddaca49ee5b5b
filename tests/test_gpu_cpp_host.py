"""GPU: the C++ host mirror (gnark names over the C-ABI) proves a synthetic key and matches the oracle byte for byte."""
import os
import subprocess
import pytest
from gpu_common import ROOT

pytestmark = pytest.mark.gpu


def test_cpp_host_mirror_prove_parity(tmp_path):
    exe = str(tmp_path / "prove_parity")
    pkg, orc = os.path.join(ROOT, "gnark-whir_amd"), os.path.join(ROOT, "oracle")
    subprocess.check_call(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "tests", "cpp", "prove_parity.cpp"), "-o", exe,
                           f"-L{pkg}", "-lmi355x_groth16", f"-L{orc}", "-lgroth16_ref", f"-Wl,-rpath,{pkg}", f"-Wl,-rpath,{orc}",
                           "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"])
    env = dict(os.environ, OMP_WAIT_POLICY="passive")   # the oracle's OpenMP barriers under a CPU quota, see oracle/cref.py
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0 and "OK 164 proof bytes identical" in out.stdout, out.stdout + out.stderr
