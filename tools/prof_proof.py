"""Per-proof kernel profile at the benchmark shape: `rocprofv3 --kernel-trace -d DIR -o p -- python3 tools/prof_proof.py [log_n] [proofs] [ranges]`
proves the bench.py workload (same seeds, committed wires removed from K, no Pedersen MSMs) `proofs` times on ONE context, one proof at a
time, inputs in HBM; `python tools/prof_proof.py --summary DIR/p_results.db [proofs]` prints the per-kernel time PER PROOF (sum of launch
durations / proofs; kernels on different streams overlap, so the column does not add up to the proof's latency)."""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SETUP = ("k_gen_", "k_xyzz_dbl_c", "k_xyzz_batch_to_affine", "k_xyzz_from_affine", "k_g1_to_rprime", "k_g2_to_rprime", "k_expand_points", "k_field_op", "k_pow_table",
         "k_tw_layout", "k_sc_layout", "k_msm2_precompute", "__amd_rocclr_fillBuffer")


def summary(db, proofs):
    import sqlite3
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), sum(end - start), min(end - start), max(end - start) from kernels group by name order by 3 desc").fetchall()
    tot = 0.0
    print(f"{'kernel':58s} {'launches/proof':>14s} {'ms/proof':>9s} {'avg us':>9s} {'max us':>9s}")
    for n, k, t, mn, mx in rows:
        short = n.split("(")[0].replace("void ", "")
        if short.startswith(SETUP):
            continue
        tot += t
        print(f"{short[:58]:58s} {k / proofs:14.1f} {t / 1e6 / proofs:9.3f} {t / k / 1e3:9.1f} {mx / 1e3:9.1f}")
    print(f"{'sum over kernels (streams overlap)':58s} {'':14s} {tot / 1e6 / proofs:9.3f}")


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--summary":
        return summary(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 12)
    sys.path.insert(0, ROOT)
    import numpy as np
    import bench
    B = bench._binding()
    log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 23
    proofs = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    N = 1 << log_n
    if len(sys.argv) > 3 and sys.argv[3] == "ranges":   # roctx ranges around the host-side phases (rocprofv3 --marker-trace --kernel-trace)
        assert B.load().mi_debug_set_trace_ranges(1) == 0
    ctx = B.Context(0)
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
    did = bench.dist_id_of(B, sys.argv[4] if len(sys.argv) > 4 else "whir")   # 4th argument: whir (default) | uniform | census
    W = ctx.gen_scalars(nb_wires, seed + 8, did); a = ctx.gen_scalars(n_constraints, seed + 9, did); b = ctx.gen_scalars(n_constraints, seed + 10, 0)
    c = ctx.alloc(32 * n_constraints); ctx.field_op_dev(0, 2, c.ptr, a.ptr, b.ptr, n_constraints)
    rs = ctx.gen_scalars(2, seed + 11, 0).download((2, 4)); ctx.sync()
    ts = []
    for k in range(proofs):
        t0 = time.perf_counter()
        ctx.prove(pkh, W.ptr, a.ptr, b.ptr, c.ptr, rs[0], rs[1], device=True, n_wires=nb_wires, n_constraints=n_constraints)
        ts.append((time.perf_counter() - t0) * 1e3)
    print("proof latencies ms:", " ".join(f"{t:.2f}" for t in ts), flush=True)
    ctx.pk_free(pkh); ctx.close()


if __name__ == "__main__":
    main()
