"""Probe: does keeping 2 proofs in flight (2 host threads, 2 contexts, ONE shared proving key) raise proofs/s?
Each context owns its streams/workspaces; the key is read-only during prove.  Usage: python tools/pipeline_probe.py [log_n] [proofs] [depth]"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib
import numpy as np
B = importlib.import_module("gnark-whir_amd.binding")

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 23
K = int(sys.argv[2]) if len(sys.argv) > 2 else 12
depth = int(sys.argv[3]) if len(sys.argv) > 3 else 2
N = 1 << log_n
ctxs = [B.Context(0) for _ in range(depth)]
ctx = ctxs[0]
seed = 0x57484952 + 1
nb_wires, nb_public, n_constraints = N - 1000, 4097, N - 100
rng = np.random.default_rng(seed)
inf_a = (rng.integers(0, 100, nb_wires) < 10).astype(np.uint8)
inf_b = (rng.integers(0, 100, nb_wires) < 50).astype(np.uint8)
na, nb, nk = int((inf_a == 0).sum()), int((inf_b == 0).sum()), nb_wires - nb_public
g1a, g1b, g1k, g1z = ctx.gen_g1(na, seed + 1), ctx.gen_g1(nb, seed + 2), ctx.gen_g1(nk, seed + 3), ctx.gen_g1(N, seed + 4)
g2b = ctx.gen_g2(nb, seed + 5)
small = ctx.gen_g1(3, seed + 6).download((3, 8)); small2 = ctx.gen_g2(2, seed + 7).download((2, 16))
pk = {"log_n": log_n, "nb_public": nb_public, "nb_wires": nb_wires,
      "g1_a": (g1a.ptr, na), "g1_b": (g1b.ptr, nb), "g1_k": (g1k.ptr, nk), "g1_z": (g1z.ptr, N), "g2_b": (g2b.ptr, nb),
      "alpha1": small[0], "beta1": small[1], "delta1": small[2], "beta2": small2[0], "delta2": small2[1],
      "infinity_a": inf_a, "infinity_b": inf_b}
pkh = ctx.pk_load(pk, device_points=True)
W = ctx.gen_scalars(nb_wires, seed + 8, 1)
a = ctx.gen_scalars(n_constraints, seed + 9, 1); b = ctx.gen_scalars(n_constraints, seed + 10, 0)
c = ctx.alloc(32 * n_constraints)
ctx.field_op_dev(0, 2, c.ptr, a.ptr, b.ptr, n_constraints)
rs = ctx.gen_scalars(2, seed + 11, 0).download((2, 4))
ctx.sync()

def prove(cx):
    return cx.prove(pkh, W.ptr, a.ptr, b.ptr, c.ptr, rs[0], rs[1], device=True, n_wires=nb_wires, n_constraints=n_constraints)

ref = prove(ctx)[0]["raw"].copy()
for cx in ctxs:
    for _ in range(2):
        assert (prove(cx)[0]["raw"] == ref).all()

# serial
t0 = time.perf_counter()
for _ in range(K):
    prove(ctx)
dt = time.perf_counter() - t0
print(f"serial   : {K / dt:7.2f} proofs/s  ({dt / K * 1e3:.2f} ms/proof)", flush=True)

# pipelined: `depth` threads, K proofs in total
bad = []
def worker(cx, n):
    for _ in range(n):
        if not (prove(cx)[0]["raw"] == ref).all():
            bad.append(1)
ths = [threading.Thread(target=worker, args=(ctxs[i], K // depth)) for i in range(depth)]
t0 = time.perf_counter()
for t in ths: t.start()
for t in ths: t.join()
dt = time.perf_counter() - t0
n = (K // depth) * depth
print(f"depth {depth}  : {n / dt:7.2f} proofs/s  ({dt / n * 1e3:.2f} ms/proof)  mismatches={len(bad)}", flush=True)
