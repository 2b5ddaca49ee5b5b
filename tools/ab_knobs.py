"""Same-box, same-process A/B of named knobs (mi_debug_set_knob) on the benchmarked workload: the key is loaded ONCE, the configurations
take turns round by round (A B C A B C ...), every proof of every configuration must give the same bytes.

    python tools/ab_knobs.py [--log-n 23] [--rounds 4] [--steps 24] [--dist whir] "l1_wg=4" "l1_wg=4,l1_waves=2" ...

The empty configuration "" (always run first) is the library's defaults.  Per configuration and round: proofs/s on the caller's path
(bench.py's step: Commit + host-input prove with the PoK, callers = in_flight + 1), proofs/s with the inputs in HBM, the single-proof
latency (inputs in HBM, nothing else on the GPU) and the level-1 launch's in-job duration.  The summary gives the mean per configuration
and, against the defaults, the number of rounds won -- the adoption rule of DESIGN.md section 8 is stated on that count."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def parse(cfg):
    out = {}
    for part in [x for x in cfg.split(",") if x]:
        k, _, v = part.partition("=")
        out[k.strip()] = int(v)
    return out


DEFAULTS = {"l1_wg": 4, "g2_wg": 1, "l1_waves": 3, "z_waves": 0, "g1_grid_per_cu": 0, "g2_grid_per_cu": 0, "count_per": 0, "plain_scatter": 0,
            "finisher": 1, "finisher_max": 0, "finisher_min_level": 2, "hold_accum": 0, "item_l1": 0, "item_l2": 0, "reduce_seg": 0, "ntt_lds_floor_kb": 0, "z_count_fused": 1, "flat_item_l1": 0, "dense_item_l1": 0}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-n", type=int, default=23)
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--in-flight", type=int, default=3)
    ap.add_argument("--dist", choices=["whir", "uniform", "census"], default="whir")
    ap.add_argument("--defaults", default="", help="knobs applied under every configuration (the baseline the others are compared with)")
    ap.add_argument("configs", nargs="*")
    args = ap.parse_args()
    B = bench._binding()
    log_n = args.log_n
    N = 1 << log_n
    pool = B.Prover(0, args.in_flight)
    ctx = pool.ctx(0)
    seed = 0x57484952 + 1
    nb_wires, nb_public, n_constraints = N - 1000, 4097, N - 100
    rng = np.random.default_rng(seed)
    inf_a = (rng.integers(0, 100, nb_wires) < 10).astype(np.uint8); inf_b = (rng.integers(0, 100, nb_wires) < 50).astype(np.uint8)
    n_committed = max(1, N >> 5)
    cp = np.sort(np.random.default_rng(seed + 77).choice(nb_wires - 1 - nb_public, n_committed, replace=False).astype(np.uint32) + np.uint32(nb_public))
    cw = np.concatenate([cp, np.array([nb_wires - 1], dtype=np.uint32)])
    na, nb = int((inf_a == 0).sum()), int((inf_b == 0).sum()); nk = nb_wires - nb_public - n_committed - 1
    g1a, g1b, g1k, g1z, g2b = ctx.gen_g1(na, seed + 1), ctx.gen_g1(nb, seed + 2), ctx.gen_g1(nk, seed + 3), ctx.gen_g1(N, seed + 4), ctx.gen_g2(nb, seed + 5)
    small = ctx.gen_g1(3, seed + 6).download((3, 8)); small2 = ctx.gen_g2(2, seed + 7).download((2, 16))
    pk = {"log_n": log_n, "nb_public": nb_public, "nb_wires": nb_wires, "g1_a": (g1a.ptr, na), "g1_b": (g1b.ptr, nb), "g1_k": (g1k.ptr, nk), "g1_z": (g1z.ptr, N),
          "g2_b": (g2b.ptr, nb), "alpha1": small[0], "beta1": small[1], "delta1": small[2], "beta2": small2[0], "delta2": small2[1],
          "infinity_a": inf_a, "infinity_b": inf_b, "committed_wires": cw}
    pkh = ctx.pk_load(pk, device_points=True)
    for d in (g1a, g1b, g1k, g2b):
        d.free()
    dist_id = bench.dist_id_of(B, args.dist)
    W = ctx.gen_scalars(nb_wires, seed + 8, dist_id); a = ctx.gen_scalars(n_constraints, seed + 9, dist_id); b = ctx.gen_scalars(n_constraints, seed + 10, 0)
    c = ctx.alloc(32 * n_constraints); ctx.field_op_dev(0, 2, c.ptr, a.ptr, b.ptr, n_constraints)
    rs = ctx.gen_scalars(3, seed + 11, 0).download((3, 4)); ctx.sync()
    Wh, ah, bh = W.download((nb_wires, 4)), a.download((n_constraints, 4)), b.download((n_constraints, 4))
    basis = ctx.gen_g1(n_committed, seed + 12).download((n_committed, 8)); sigma = ctx.gen_g1(n_committed, seed + 13).download((n_committed, 8))
    ped = ctx.pedersen_pk_load(basis, sigma); values = np.ascontiguousarray(Wh[cp])

    def apply(cfg):
        kn = dict(DEFAULTS); kn.update(parse(args.defaults)); kn.update(parse(cfg))
        for k, v in kn.items():
            pool.set_knob(k, v)

    def step():
        cm = pool.commit(ped, values)
        pr, st = pool.wait(pool.submit_bsb22(pkh, Wh, ah, bh, None, rs[0], rs[1], [(ped, values)], rs[2]))
        return B.proof_write(pr["raw"], cm.reshape(1, 8), pr["pok"]), st

    def dev():
        return pool.submit(pkh, W.ptr, a.ptr, b.ptr, c.ptr, rs[0], rs[1], device=True, n_wires=nb_wires, n_constraints=n_constraints)

    from concurrent.futures import ThreadPoolExecutor
    callers = pool.in_flight + 1
    ex = ThreadPoolExecutor(callers)
    configs = [""] + [x for x in args.configs if x != ""]
    ref = ref_body = None
    res = {cfg: [] for cfg in configs}
    for rnd in range(args.rounds + 1):   # round 0 = warm-up of every configuration (workspaces), not recorded
        for cfg in configs:
            apply(cfg)
            for i in range(pool.in_flight):   # single-proof latency, each context once (the last one counts)
                t0 = time.perf_counter()
                pr, _ = pool.ctx(i).prove(pkh, W.ptr, a.ptr, b.ptr, c.ptr, rs[0], rs[1], device=True, n_wires=nb_wires, n_constraints=n_constraints)
                lat = (time.perf_counter() - t0) * 1e3
                body = B.proof_write(pr["raw"])
                ref_body = ref_body or body
                assert body == ref_body, f"proof bytes differ under [{cfg}]"
            list(ex.map(lambda _: step(), range(callers)))
            ctx.sync(); t0 = time.perf_counter()
            done = list(ex.map(lambda _: step(), range(args.steps)))
            ctx.sync(); host_rate = args.steps / (time.perf_counter() - t0)
            for bts, _ in done:
                ref = ref or bts
                assert bts == ref, f"196-byte proofs differ under [{cfg}]"
            acc_ms = sum(st["g1_accum_kernel_ms"] for _, st in done) / max(1, sum(st["g1_accum_launches"] for _, st in done))
            for t in [dev() for _ in range(pool.in_flight)]:
                pool.wait(t)
            ctx.sync(); t0 = time.perf_counter()
            outs = [pool.wait(t)[0]["raw"] for t in [dev() for _ in range(args.steps)]]
            ctx.sync(); dev_rate = args.steps / (time.perf_counter() - t0)
            for raw in outs:
                assert B.proof_write(raw) == ref_body, f"device-input proofs differ under [{cfg}]"
            if rnd:
                res[cfg].append((host_rate, dev_rate, lat, acc_ms))
                print(f"r{rnd} [{cfg}] value {host_rate:.2f} hbm {dev_rate:.2f} lat {lat:.2f} accum launch {acc_ms:.2f} ms", flush=True)
    base = res[""]
    print("---- summary (mean over rounds; wins = rounds in which the configuration beat the defaults of the SAME round)")
    for cfg in configs:
        r = np.array(res[cfg])
        wins_v = sum(1 for x, y in zip(res[cfg], base) if x[0] > y[0]); wins_h = sum(1 for x, y in zip(res[cfg], base) if x[1] > y[1])
        print(f"[{cfg or 'defaults'}] value {r[:, 0].mean():.2f} ({(r[:, 0].mean() / np.array(base)[:, 0].mean() - 1) * 100:+.1f} %, wins {wins_v}/{len(base)})  "
              f"hbm {r[:, 1].mean():.2f} ({(r[:, 1].mean() / np.array(base)[:, 1].mean() - 1) * 100:+.1f} %, wins {wins_h}/{len(base)})  "
              f"lat {r[:, 2].mean():.2f} ms  accum launch {r[:, 3].mean():.2f} ms", flush=True)
    pool.close()


if __name__ == "__main__":
    main()
