"""Soak: many proofs through the prover pool -- device-resident and HOST inputs mixed (the upload stage) -- every one compared
with the proof a plain context computed alone.   usage: python tools/soak_pool.py [log_n] [jobs] [in_flight]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
from gpu_common import load_binding
B = load_binding()
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
jobs = int(sys.argv[2]) if len(sys.argv) > 2 else 600
depth = int(sys.argv[3]) if len(sys.argv) > 3 else 3
N = 1 << log_n
nb_wires, nb_public, n_constraints = N - 1000, 4097, N - 100
pool = B.Prover(0, depth); c0 = pool.ctx(0)
rng = np.random.default_rng(5)
inf_a = (rng.integers(0, 100, nb_wires) < 10).astype(np.uint8); inf_b = (rng.integers(0, 100, nb_wires) < 50).astype(np.uint8)
na, nb, nk = int((inf_a == 0).sum()), int((inf_b == 0).sum()), nb_wires - nb_public
g1a, g1b, g1k, g1z, g2b = c0.gen_g1(na, 1), c0.gen_g1(nb, 2), c0.gen_g1(nk, 3), c0.gen_g1(N, 4), c0.gen_g2(nb, 5)
small = c0.gen_g1(3, 6).download((3, 8)); small2 = c0.gen_g2(2, 7).download((2, 16))
pk = {"log_n": log_n, "nb_public": nb_public, "nb_wires": nb_wires, "g1_a": (g1a.ptr, na), "g1_b": (g1b.ptr, nb), "g1_k": (g1k.ptr, nk),
      "g1_z": (g1z.ptr, N), "g2_b": (g2b.ptr, nb), "alpha1": small[0], "beta1": small[1], "delta1": small[2], "beta2": small2[0],
      "delta2": small2[1], "infinity_a": inf_a, "infinity_b": inf_b}
pkh = c0.pk_load(pk, device_points=True)
wit = []
for w in range(2):
    W = c0.gen_scalars(nb_wires, 100 + w, 1 - w)
    a = c0.gen_scalars(n_constraints, 200 + w, 1); b = c0.gen_scalars(n_constraints, 300 + w, 0)
    c = c0.alloc(32 * n_constraints); c0.field_op_dev(0, 2, c.ptr, a.ptr, b.ptr, n_constraints)
    wit.append((W, a, b, c))
rs = c0.gen_scalars(8, 400, 0).download((8, 4))
hwit = [tuple(x.download((n, 4)) for x, n in zip(wt, (nb_wires, n_constraints, n_constraints, n_constraints))) for wt in wit]   # host copies
c0.sync()
single = B.Context(0)
combos = [(w, k) for w in range(2) for k in range(4)]
ref = {}
for w, k in combos:
    ref[(w, k)] = single.prove(pkh, *(x.ptr for x in wit[w]), rs[2 * (k % 4)], rs[2 * (k % 4) + 1], device=True, n_wires=nb_wires, n_constraints=n_constraints)[0]["raw"].copy()
single.close()
t0 = time.perf_counter(); bad = 0; done = 0
window = []
for j in range(jobs):
    w, k = combos[j % len(combos)]
    if j % 3 == 1:   # every third job hands over host pointers
        window.append(((w, k), pool.submit(pkh, *hwit[w], rs[2 * k], rs[2 * k + 1])))
    else:
        window.append(((w, k), pool.submit(pkh, *(x.ptr for x in wit[w]), rs[2 * k], rs[2 * k + 1], device=True, n_wires=nb_wires, n_constraints=n_constraints)))
    if len(window) >= 4 * depth:
        key, t = window.pop(0)
        bad += not np.array_equal(pool.wait(t)[0]["raw"], ref[key]); done += 1
        if done % 100 == 0:
            print(f"{done} proofs, {bad} mismatches, {done / (time.perf_counter() - t0):.1f} proofs/s", flush=True)
for key, t in window:
    bad += not np.array_equal(pool.wait(t)[0]["raw"], ref[key]); done += 1
print(f"SOAK log_n={log_n} in_flight={depth}: {done} proofs, {bad} mismatches, {done / (time.perf_counter() - t0):.1f} proofs/s", flush=True)
pool.close()
sys.exit(1 if bad else 0)
