"""Shared by bench.py, tools/*.py and tools/benchlib: where the repo is, how the ctypes binding is loaded, which scalar mixes exist."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
BENCH_PY = os.path.join(ROOT, "bench.py")


def _binding():
    spec = importlib.util.spec_from_file_location("gnark_whir_amd_binding", os.path.join(ROOT, "gnark-whir_amd", "binding.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["gnark_whir_amd_binding"] = mod
    spec.loader.exec_module(mod)
    return mod


def _sha16(path):
    import hashlib
    try:
        return hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]
    except OSError:
        return None


def dist_id_of(B, name):
    """the `dist` argument of mi_gen_scalars_dev / ref_gen_scalars for a named witness mix: `whir` = BASELINE.md 3's 45 / 25 / 5 / 25 guess,
    `uniform`, `census` = the midpoint mix tools/wire_census.py derives from the reference's circuit (profiles/r06_wire_census.txt)"""
    if name == "uniform":
        return B.DIST_UNIFORM
    if name.startswith("mix:"):   # a user's own census (tools/wire_census.py --params ...): per-mille of bits, bytes, 64-bit values
        return B.dist_mix(*[int(x) for x in name[4:].split(",")])
    if name == "census":
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import wire_census
        return B.dist_mix(*wire_census.census_mix_permille())
    return B.DIST_WHIR
