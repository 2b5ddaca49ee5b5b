"""Latency of the mid-solve BSB22 commitment (mi_pedersen_commit): host values, device-resident basis."""
import sys, time
import numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, "oracle")
from gpu_common import load_binding
import cref
B = load_binding(); ctx = B.Context(0)
for n in (1 << 10, 1 << 14, 1 << 16, 1 << 18, 1 << 14):
    basis = cref.gen_g1(n, 1); vals = cref.gen_scalars(n, 2, 1)
    pk = ctx.pedersen_pk_load(basis, basis)
    ts = []
    for _ in range(6):
        t = time.perf_counter(); ctx.pedersen_commit(pk, vals); ts.append((time.perf_counter() - t) * 1e3)
    print(n, " ".join(f"{x:.2f}" for x in ts), flush=True)
    ctx.pedersen_pk_free(pk)
