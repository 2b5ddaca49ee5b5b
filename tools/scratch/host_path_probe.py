"""Host-pointer prove (the cgo path): PCIe-inclusive time per proof at N = 2^23 vs the device-resident path."""
import sys, time
import numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, "oracle")
from gpu_common import load_binding
B = load_binding(); ctx = B.Context(0)
log_n = 23; N = 1 << log_n
nb_wires, nb_public, n_constraints = N - 1000, 4097, N - 100
rng = np.random.default_rng(1)
inf_a = (rng.integers(0, 100, nb_wires) < 10).astype(np.uint8); inf_b = (rng.integers(0, 100, nb_wires) < 50).astype(np.uint8)
na, nb, nk = int((inf_a == 0).sum()), int((inf_b == 0).sum()), nb_wires - nb_public
g1a, g1b, g1k, g1z, g2b = ctx.gen_g1(na, 1), ctx.gen_g1(nb, 2), ctx.gen_g1(nk, 3), ctx.gen_g1(N, 4), ctx.gen_g2(nb, 5)
small = ctx.gen_g1(3, 6).download((3, 8)); small2 = ctx.gen_g2(2, 7).download((2, 16))
pk = {"log_n": log_n, "nb_public": nb_public, "nb_wires": nb_wires, "g1_a": (g1a.ptr, na), "g1_b": (g1b.ptr, nb), "g1_k": (g1k.ptr, nk),
      "g1_z": (g1z.ptr, N), "g2_b": (g2b.ptr, nb), "alpha1": small[0], "beta1": small[1], "delta1": small[2], "beta2": small2[0],
      "delta2": small2[1], "infinity_a": inf_a, "infinity_b": inf_b}
pkh = ctx.pk_load(pk, device_points=True)
Wd = ctx.gen_scalars(nb_wires, 8, 1); ad = ctx.gen_scalars(n_constraints, 9, 1); bd = ctx.gen_scalars(n_constraints, 10, 0)
cd = ctx.alloc(32 * n_constraints); ctx.field_op_dev(0, 2, cd.ptr, ad.ptr, bd.ptr, n_constraints)
W, a, b, c = Wd.download((nb_wires, 4)), ad.download((n_constraints, 4)), bd.download((n_constraints, 4)), cd.download((n_constraints, 4))
rs = ctx.gen_scalars(2, 11, 0).download((2, 4))
p_dev, _ = ctx.prove(pkh, Wd.ptr, ad.ptr, bd.ptr, cd.ptr, rs[0], rs[1], device=True, n_wires=nb_wires, n_constraints=n_constraints)
for rep in range(4):
    t = time.perf_counter(); p_host, st = ctx.prove(pkh, W, a, b, c, rs[0], rs[1]); dt = (time.perf_counter() - t) * 1e3
    print(f"host-pointer prove: {dt:.1f} ms wall (h2d span {st['h2d_ms']:.1f} ms, total_ms {st['total_ms']:.1f})", flush=True)
for rep in range(3):
    t = time.perf_counter(); ctx.prove(pkh, Wd.ptr, ad.ptr, bd.ptr, cd.ptr, rs[0], rs[1], device=True, n_wires=nb_wires, n_constraints=n_constraints); print(f"device-pointer prove: {(time.perf_counter() - t) * 1e3:.1f} ms", flush=True)
assert np.array_equal(p_host["raw"], p_dev["raw"])
print("host-path proof == device-path proof")
# the same through the prover pool, host buffers, D proofs in flight
for D in (2, 3):
    pool = B.Prover(0, D)
    for i in range(D):
        pool.ctx(i).prove(pkh, W, a, b, c, rs[0], rs[1])
    K = 12
    t = time.perf_counter()
    tk = [pool.submit(pkh, W, a, b, c, rs[0], rs[1]) for _ in range(K)]
    res = [pool.wait(x) for x in tk]
    dt = time.perf_counter() - t
    assert all(np.array_equal(r[0]["raw"], p_dev["raw"]) for r in res)
    print(f"pool, host buffers, {D} in flight: {dt / K * 1e3:.1f} ms per proof = {K / dt:.2f} proofs/s", flush=True)
    pool.close()
