"""Soak: many proofs through the prover pool -- device-resident inputs, HOST inputs (the upload stage; with c and with c = NULL) and host
inputs with a BSB22 commitment (mi_prover_commit + mi_prover_submit_bsb22: the PoK MSM enqueued from a helper thread beside the proof)
mixed -- every one compared with the proof a plain context computed alone, every commitment and PoK with the first one computed.
Half the witnesses are UNIFORM wire values (13-15 digits per scalar in the four wire MSMs instead of the WHIR mix's ~4.4), so the rate
printed here is well below bench.py's (22.8 against 31 proofs/s at 2^23).
usage: python tools/soak_pool.py [log_n] [jobs] [in_flight] [knobs: name=value,...]"""
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
for part in [x for x in (sys.argv[4] if len(sys.argv) > 4 else "").split(",") if x]:   # mi_debug_set_knob on every context of the pool
    k_, _, v_ = part.partition("=")
    pool.set_knob(k_.strip(), int(v_))
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
n_ped = max(1, N >> 5)
basis = c0.gen_g1(n_ped, 500).download((n_ped, 8)); sigma = c0.gen_g1(n_ped, 501).download((n_ped, 8))
ped = c0.pedersen_pk_load(basis, sigma)
vals = [np.ascontiguousarray(hwit[w][0][:n_ped]) for w in range(2)]
cm_ref = [pool.commit(ped, v) for v in vals]
pok_ref = [None, None]
t0 = time.perf_counter(); bad = 0; done = 0
window = []


def check(key, t, kind):
    global bad, done
    pr = pool.wait(t)[0]
    ok = np.array_equal(pr["raw"], ref[key])
    if kind == "bsb":
        w = key[0]
        if pok_ref[w] is None:
            pok_ref[w] = pr["pok"].copy()
        ok = ok and np.array_equal(pr["pok"], pok_ref[w])
    bad += not ok; done += 1
    if done % 100 == 0:
        print(f"{done} proofs, {bad} mismatches, {done / (time.perf_counter() - t0):.1f} proofs/s", flush=True)


for j in range(jobs):
    w, k = combos[j % len(combos)]
    r, s = rs[2 * k], rs[2 * k + 1]
    if j % 5 == 1:     # host pointers, c given
        window.append(((w, k), pool.submit(pkh, *hwit[w], r, s), "host"))
    elif j % 5 == 2:   # host pointers, c formed on the device
        window.append(((w, k), pool.submit(pkh, hwit[w][0], hwit[w][1], hwit[w][2], None, r, s), "host"))
    elif j % 5 == 3:   # the whole caller's step: Commit, then the proof with its PoK
        bad += not np.array_equal(pool.commit(ped, vals[w]), cm_ref[w])
        window.append(((w, k), pool.submit_bsb22(pkh, hwit[w][0], hwit[w][1], hwit[w][2], None, r, s, [(ped, vals[w])], rs[0]), "bsb"))
    else:
        window.append(((w, k), pool.submit(pkh, *(x.ptr for x in wit[w]), r, s, device=True, n_wires=nb_wires, n_constraints=n_constraints), "dev"))
    if len(window) >= 4 * depth:
        check(*window.pop(0))
for item in window:
    check(*item)
print(f"SOAK log_n={log_n} in_flight={depth}: {done} proofs, {bad} mismatches, {done / (time.perf_counter() - t0):.1f} proofs/s", flush=True)
pool.close()
sys.exit(1 if bad else 0)
