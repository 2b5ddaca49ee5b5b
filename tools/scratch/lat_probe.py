"""single-proof latency of the host-input paths at N = 2^23, phase by phase (one-off probe)"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
B = bench._binding()
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 23
N = 1 << log_n
pool = B.Prover(0, 3); ctx = pool.ctx(0)
seed = 0x57484952 + 1
nb_wires, nb_public, n_constraints = N - 1000, 4097, N - 100
rng = np.random.default_rng(seed)
inf_a = (rng.integers(0, 100, nb_wires) < 10).astype(np.uint8); inf_b = (rng.integers(0, 100, nb_wires) < 50).astype(np.uint8)
n_committed = N >> 5
cp = np.sort(np.random.default_rng(seed + 77).choice(nb_wires - 1 - nb_public, n_committed, replace=False).astype(np.uint32) + np.uint32(nb_public))
cw = np.concatenate([cp, np.array([nb_wires - 1], dtype=np.uint32)])
na, nb = int((inf_a == 0).sum()), int((inf_b == 0).sum()); nk = nb_wires - nb_public - n_committed - 1
g1a, g1b, g1k, g1z, g2b = ctx.gen_g1(na, seed + 1), ctx.gen_g1(nb, seed + 2), ctx.gen_g1(nk, seed + 3), ctx.gen_g1(N, seed + 4), ctx.gen_g2(nb, seed + 5)
small = ctx.gen_g1(3, seed + 6).download((3, 8)); small2 = ctx.gen_g2(2, seed + 7).download((2, 16))
pk = {"log_n": log_n, "nb_public": nb_public, "nb_wires": nb_wires, "g1_a": (g1a.ptr, na), "g1_b": (g1b.ptr, nb), "g1_k": (g1k.ptr, nk), "g1_z": (g1z.ptr, N), "g2_b": (g2b.ptr, nb),
      "alpha1": small[0], "beta1": small[1], "delta1": small[2], "beta2": small2[0], "delta2": small2[1], "infinity_a": inf_a, "infinity_b": inf_b, "committed_wires": cw}
pkh = ctx.pk_load(pk, device_points=True)
W = ctx.gen_scalars(nb_wires, seed + 8, 1); a = ctx.gen_scalars(n_constraints, seed + 9, 1); b = ctx.gen_scalars(n_constraints, seed + 10, 0)
c = ctx.alloc(32 * n_constraints); ctx.field_op_dev(0, 2, c.ptr, a.ptr, b.ptr, n_constraints)
rs = ctx.gen_scalars(3, seed + 11, 0).download((3, 4)); ctx.sync()
Wh, ah, bh, ch = W.download((nb_wires, 4)), a.download((n_constraints, 4)), b.download((n_constraints, 4)), c.download((n_constraints, 4))
basis = ctx.gen_g1(n_committed, seed + 12).download((n_committed, 8)); sigma = ctx.gen_g1(n_committed, seed + 13).download((n_committed, 8))
ped = ctx.pedersen_pk_load(basis, sigma); values = np.ascontiguousarray(Wh[cp])
for i in range(3):
    pool.ctx(i).prove(pkh, W.ptr, a.ptr, b.ptr, c.ptr, rs[0], rs[1], device=True, n_wires=nb_wires, n_constraints=n_constraints)
keys = ("h2d_ms", "compute_h_ms", "msm_a_ms", "msm_b1_ms", "msm_b2_ms", "msm_k_ms", "msm_z_ms", "filter_ms", "assemble_ms", "total_ms")
def show(name, fn, reps=4):
    for k in range(reps):
        t0 = time.perf_counter(); st = fn(); dt = (time.perf_counter() - t0) * 1e3
    print(f"{name:44s} {dt:7.2f} ms  " + " ".join(f"{k[:-3]}={st[k]:.1f}" for k in keys), flush=True)
show("device inputs, ctx0.prove", lambda: pool.ctx(0).prove(pkh, W.ptr, a.ptr, b.ptr, c.ptr, rs[0], rs[1], device=True, n_wires=nb_wires, n_constraints=n_constraints)[1])
show("device inputs, pool", lambda: pool.wait(pool.submit(pkh, W.ptr, a.ptr, b.ptr, c.ptr, rs[0], rs[1], device=True, n_wires=nb_wires, n_constraints=n_constraints))[1])
show("host W a b c, ctx0.prove (direct)", lambda: pool.ctx(0).prove(pkh, Wh, ah, bh, ch, rs[0], rs[1])[1])
show("host W a b, ctx0.prove (direct, c derived)", lambda: pool.ctx(0).prove(pkh, Wh, ah, bh, None, rs[0], rs[1])[1])
show("host W a b c, pool", lambda: pool.wait(pool.submit(pkh, Wh, ah, bh, ch, rs[0], rs[1]))[1])
show("host W a b, pool (c derived)", lambda: pool.wait(pool.submit(pkh, Wh, ah, bh, None, rs[0], rs[1]))[1])
show("host W a b, pool bsb22 (no commit)", lambda: pool.wait(pool.submit_bsb22(pkh, Wh, ah, bh, None, rs[0], rs[1], [(ped, values)], rs[2]))[1])
def full():
    pool.commit(ped, values)
    return pool.wait(pool.submit_bsb22(pkh, Wh, ah, bh, None, rs[0], rs[1], [(ped, values)], rs[2]))[1]
show("commit + host W a b, pool bsb22", full)
# ---- throughput: where do the commitment's 3.5 % go?  20 steps from 4 caller threads, variants
from concurrent.futures import ThreadPoolExecutor
ex = ThreadPoolExecutor(4)
def tput(name, step, reps=3):
    best = 0
    for _ in range(reps):
        list(ex.map(lambda _: step(), range(4)))
        ctx.sync(); t0 = time.perf_counter()
        list(ex.map(lambda _: step(), range(24)))
        ctx.sync(); best = max(best, 24 / (time.perf_counter() - t0))
    print(f"throughput {name:52s} {best:6.2f} proofs/s", flush=True)
plain = lambda: pool.wait(pool.submit(pkh, Wh, ah, bh, None, rs[0], rs[1]))
bsb = lambda: pool.wait(pool.submit_bsb22(pkh, Wh, ah, bh, None, rs[0], rs[1], [(ped, values)], rs[2]))
def commit_plain():
    pool.commit(ped, values); return plain()
def commit_bsb():
    pool.commit(ped, values); return bsb()
dev = lambda: pool.wait(pool.submit(pkh, W.ptr, a.ptr, b.ptr, c.ptr, rs[0], rs[1], device=True, n_wires=nb_wires, n_constraints=n_constraints))
for _ in range(2):
    tput("device inputs, plain", dev)
    tput("host inputs, plain submit", plain)
    tput("host inputs, submit_bsb22 (PoK in the job)", bsb)
    tput("host inputs, Commit + plain submit", commit_plain)
    tput("host inputs, Commit + submit_bsb22 (the bench step)", commit_bsb)
