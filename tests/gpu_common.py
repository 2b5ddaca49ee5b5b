"""Loads the product binding (gnark-whir_amd/binding.py) for the GPU tests.  No fallback: a
missing library or device is a hard failure, never a skip to a CPU path."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_binding():
    name = "gnark_whir_amd_binding"
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "gnark-whir_amd", "binding.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod
