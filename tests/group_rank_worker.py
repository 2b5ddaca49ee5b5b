"""ONE RANK of a multi-process device group (mi_group_create_rank_ex, csrc/group.hip) -- the process tests/rank_launcher.py starts
`world` times for tests/test_gpu_group_multiprocess.py.  All ranks use device 0 and meet through the host-staged transport
(MI_GROUP_TRANSPORT_HOST): RCCL refuses two ranks on one device, everything else of the one-rank-per-process flow -- rank-local
indexing with rank0 != 0, the lead / non-lead split of the prove, the agreements, the all-gathers, the h-slice and bucket-slice
batches with REMOTE ranks -- is the code an 8-GPU node runs (BASELINE configs[4]).  Every rank checks its own results
against the oracle and prints one JSON line.

    python group_rank_worker.py <scenario> <uid hex>        (MI_RANK, MI_WORLD in the environment)"""
import hashlib
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, os.path.join(ROOT, "oracle"), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np   # noqa: E402
import cref          # noqa: E402
from helpers import synthetic_pk   # noqa: E402
from gpu_common import load_binding   # noqa: E402


def uid_of(seed_hex, k):
    h = hashlib.sha512((seed_hex + ":" + str(k)).encode()).digest()
    return (h + h)[:128]


def slices_of(B, c, pk, world, rank, keep, g=None):
    """device copies of rank `rank`'s slices of the five point arrays (what mi_pk_load_sharded_dev adopts)"""
    N = 1 << pk["log_n"]
    ia, ib = np.asarray(pk["infinity_a"]), np.asarray(pk["infinity_b"])
    cw = set(int(x) for x in (pk.get("committed_wires") if pk.get("committed_wires") is not None else []))
    in_k = np.array([j >= pk["nb_public"] and j not in cw for j in range(pk["nb_wires"])])
    ca, cb, ck = np.concatenate([[0], np.cumsum(ia == 0)]), np.concatenate([[0], np.cumsum(ib == 0)]), np.concatenate([[0], np.cumsum(in_k)])
    lo, hi = g.wire_range(pk["nb_wires"], rank) if g is not None else B.shard_range(pk["nb_wires"], world, rank)   # wires by the group's lead share
    zlo, zhi = B.shard_range(N - 1, world, rank)
    sl = {}
    for name, arr in (("g1_a", pk["g1_a"][ca[lo]:ca[hi]]), ("g1_b", pk["g1_b"][cb[lo]:cb[hi]]), ("g1_k", pk["g1_k"][ck[lo]:ck[hi]]),
                      ("g1_z", pk["g1_z"][zlo:zhi]), ("g2_b", pk["g2_b"][cb[lo]:cb[hi]])):
        d = c.to_dev(arr); keep.append(d); sl[name] = (d.ptr, arr.shape[0])
    return sl


def workload(log_n, seed, n_committed=0):
    N = 1 << log_n
    nb_wires, n_constraints = N - 50, N - 10
    pk = synthetic_pk(log_n, nb_wires, min(300, nb_wires // 4), seed, n_committed=n_committed)
    W = cref.gen_scalars(nb_wires, seed + 11, 1)
    a = cref.gen_scalars(n_constraints, seed + 12, 1); b = cref.gen_scalars(n_constraints, seed + 13, 0); c = cref.field_op(0, 2, a, b)
    r, s = cref.gen_scalars(2, seed + 14, 0)
    want = cref.proof_write(cref.prove(pk, W, a, b, c, r, s)["raw"])
    return pk, W, a, b, c, r, s, want


def scenario_parity(B, rank, world, seed_hex):
    checks = []
    g = B.Group.rank(0, rank, world, uid_of(seed_hex, 0), transport=3)
    try:
        assert g.transport() == "host-staged" and g.world == world and g.n_local == 1 and g.rank_index() == rank
        g.exchange_selftest(4096)
        g.exchange_selftest((3 << 20) + 5)   # several chunks of the rings, a ragged tail
        checks.append("selftest")
        c = g.ctx(0)
        # ---- one MSM point-sharded over the processes, pairs resident on every rank (mi_msm_g1/g2_sharded_dev), both modes
        n, n2 = 40000, 2500
        pts = cref.gen_g1(n, 31); sc = cref.gen_scalars(n, 32, 1)
        p2 = cref.gen_g2(n2, 33); s2 = cref.gen_scalars(n2, 34, 0)
        lo, hi = B.shard_range(n, world, rank); lo2, hi2 = B.shard_range(n2, world, rank)
        d = [c.to_dev(pts[lo:hi]), c.to_dev(sc[lo:hi]), c.to_dev(p2[lo2:hi2]), c.to_dev(s2[lo2:hi2])]
        w1, w2 = cref.msm_g1(pts, sc), cref.msm_g2(p2, s2)
        for mode in (0, 1):
            assert np.array_equal(g.msm_dev([d[0].ptr], [d[1].ptr], [hi - lo], n, mode=mode), w1), f"G1 MSM mode {mode}"
            assert np.array_equal(g.msm_dev([d[2].ptr], [d[3].ptr], [hi2 - lo2], n2, mode=mode, g2=True), w2), f"G2 MSM mode {mode}"
        for x in d:
            x.free()
        checks.append("msm_g1_g2_both_modes")
        # ---- one proof at N = 2^16 point-sharded over the processes: host arrays (every process passes the whole key and the whole W,
        #      only the lead passes a, b, c) and device slices + device inputs; both modes; c given and c = None (formed on the device)
        pk, W, a, b, cc, r, s, want = workload(16, 9100, n_committed=23)
        lead = rank == 0
        spk = g.pk_load(pk)
        for mode in (0, 1):
            got, _ = g.prove(spk, W, a if lead else None, b if lead else None, cc if lead else None, r, s, mode=mode)
            assert B.proof_write(got["raw"]) == want, f"sharded prove (host inputs) mode {mode}"
        got, _ = g.prove(spk, W, a if lead else None, b if lead else None, None, r, s, mode=1)
        assert B.proof_write(got["raw"]) == want, "sharded prove with c = a o b formed on the device"
        g.pk_free(spk)
        keep = []
        spk = g.pk_load_dev(pk, [slices_of(B, c, pk, world, rank, keep, g)])
        wlo, whi = g.wire_range(pk["nb_wires"], rank)
        dW = c.to_dev(W[wlo:whi]); keep.append(dW)
        da = db = dc = None
        if lead:
            da, db, dc = c.to_dev(a), c.to_dev(b), c.to_dev(cc); keep += [da, db, dc]
        ptr = lambda x: None if x is None else x.ptr
        for mode in (0, 1):
            got, _ = g.prove_dev(spk, [dW.ptr], pk["nb_wires"], ptr(da), ptr(db), ptr(dc), a.shape[0], r, s, mode=mode)
            assert B.proof_write(got["raw"]) == want, f"sharded prove (device slices) mode {mode}"
        g.pk_free(spk)
        for x in keep:
            x.free()
        checks.append("prove_2p16_host_and_device_both_modes")
        # ---- the lead's share of the wires (mi_group_set_lead_share): 0 (its wire MSMs are empty), half, the even cut -- same bytes in both
        #      modes, host arrays and device slices; a rank that sets another share than its peers fails the load on EVERY rank
        pk, W, a, b, cc, r, s, want = workload(13, 9150, n_committed=5)
        for share in (0, 500, 1000):
            g.set_lead_share(share)
            spk = g.pk_load(pk)
            for mode in (0, 1):
                got, _ = g.prove(spk, W, a if lead else None, b if lead else None, cc if lead else None, r, s, mode=mode)
                assert B.proof_write(got["raw"]) == want, f"lead share {share} mode {mode} (host arrays)"
            g.pk_free(spk)
            keep = []
            spk = g.pk_load_dev(pk, [slices_of(B, c, pk, world, rank, keep, g)])
            wlo, whi = g.wire_range(pk["nb_wires"], rank)
            dW = c.to_dev(W[wlo:whi]); keep.append(dW)
            da = db = None
            if lead:
                da, db = c.to_dev(a), c.to_dev(b); keep += [da, db]
            got, _ = g.prove_dev(spk, [dW.ptr], pk["nb_wires"], ptr(da), ptr(db), None, a.shape[0], r, s, mode=1)
            assert B.proof_write(got["raw"]) == want, f"lead share {share} (device slices)"
            g.pk_free(spk)
            for x in keep:
                x.free()
        g.set_lead_share(1000 if rank == 0 else 250)
        refused = False
        try:
            g.pk_load(pk)
        except B.MiError:
            refused = True
        assert refused, "ranks that disagree on the lead share must all be refused the key"
        g.set_lead_share(0xFFFFFFFF)
        checks.append("lead_share_0_500_1000_and_disagreement")
        # ---- computeH over the ranks (worlds 2 and 4; world 3 falls back to the lead's computeH): every process passes a and b, the h
        #      slices are born where the Z pairs live; host arrays in both modes, row slices on the device with c formed there
        g.set_sharded_compute_h(True)
        spk = g.pk_load(pk)
        for mode in (0, 1):
            got, _ = g.prove(spk, W, a, b, cc if mode else None, r, s, mode=mode)
            assert B.proof_write(got["raw"]) == want, f"computeH over the ranks, host arrays, mode {mode}"
        g.pk_free(spk)
        if world in (2, 4):
            N = 1 << pk["log_n"]; M = N // world; ncs = a.shape[0]
            keep = []
            spk = g.pk_load_dev(pk, [slices_of(B, c, pk, world, rank, keep, g)])
            wlo, whi = g.wire_range(pk["nb_wires"], rank)
            lo, hi = min(rank * M, ncs), min((rank + 1) * M, ncs)
            dW, da, db = c.to_dev(W[wlo:whi]), c.to_dev(a[lo:hi]), c.to_dev(b[lo:hi]); keep += [dW, da, db]
            got, _ = g.prove_slices_dev(spk, [dW.ptr], pk["nb_wires"], [da.ptr], [db.ptr], None, ncs, r, s, mode=1)
            assert B.proof_write(got["raw"]) == want, "computeH over the ranks, row slices"
            # ... and computeH alone, this rank's slice of h against the oracle's
            dh = c.alloc(32 * M); keep.append(dh)
            g.compute_h_sharded_dev(pk["log_n"], [da.ptr], [db.ptr], None, ncs, [dh.ptr])
            assert np.array_equal(dh.download((M, 4)), cref.compute_h(pk["log_n"], a, b, cc)[rank * M:(rank + 1) * M]), "h slice"
            g.pk_free(spk)
            for x in keep:
                x.free()
        g.set_sharded_compute_h(False)
        checks.append("compute_h_over_the_ranks")
    finally:
        g.close()
    return {"ok": True, "checks": checks}


def scenario_inject(B, rank, world, seed_hex):
    """A LOCAL failure on one rank (the n-th checked HIP call of that process fails) must come back as an error on EVERY rank, quickly,
    and must not break anything: the group is recreated afterwards and proves the same bytes as before."""
    pk, W, a, b, cc, r, s, want = workload(13, 9200)
    lead = rank == 0
    lib = B.load()
    trials = []
    gen = 0

    def fresh():
        nonlocal gen
        g = B.Group.rank(0, rank, world, uid_of(seed_hex, gen), transport=3)
        gen += 1
        return g, g.pk_load(pk)

    g, spk = fresh()
    got, _ = g.prove(spk, W, a if lead else None, b if lead else None, cc if lead else None, r, s, mode=1)
    assert B.proof_write(got["raw"]) == want
    # (victim rank, n-th checked call from the start of the prove, mode): early calls = uploads and enqueues of the local phase, later
    # ones = the sorts, the bucket exchange's preparation, the collection
    plan = [(1, 1, 0), (0, 1, 0), (1, 2, 1), (0, 4, 1), (1, 7, 1), (1, 12, 0), (0, 20, 1), (1, 40, 1)]
    for victim, nth, mode in plan:
        if rank == victim:
            assert lib.mi_debug_inject_hip_failure(nth) == 0
        t0 = time.time()
        failed, msg = False, ""
        try:
            got, _ = g.prove(spk, W, a if lead else None, b if lead else None, cc if lead else None, r, s, mode=mode)
        except B.MiError as e:
            failed, msg = True, str(e)
        dt = time.time() - t0
        lib.mi_debug_inject_hip_failure(0)
        trials.append({"victim": victim, "nth": nth, "mode": mode, "failed": failed, "seconds": round(dt, 3), "msg": msg[:160]})
        if not failed:   # the injected call number lies beyond this call's checked calls on the victim: then EVERY rank must have succeeded
            assert B.proof_write(got["raw"]) == want
        # whatever happened, the ranks start over together (a failure inside an exchange marks the group broken on purpose)
        try:
            g.pk_free(spk)
        except B.MiError:
            pass
        g.close()
        g, spk = fresh()
        got, _ = g.prove(spk, W, a if lead else None, b if lead else None, cc if lead else None, r, s, mode=mode)
        assert B.proof_write(got["raw"]) == want, "the group must prove correctly after a failed call"
    # the same with computeH OVER THE RANKS (local failures are carried through its nine all-to-alls: nobody may be left waiting in one)
    for victim, nth in ((1, 3), (0, 6), (1, 10), (0, 16), (1, 24), (1, 36), (0, 50), (1, 70)):
        g.set_sharded_compute_h(True)
        if rank == victim:
            assert lib.mi_debug_inject_hip_failure(nth) == 0
        t0 = time.time()
        failed, msg = False, ""
        try:
            got, _ = g.prove(spk, W, a, b, None, r, s, mode=nth & 1)
        except B.MiError as e:
            failed, msg = True, str(e)
        dt = time.time() - t0
        lib.mi_debug_inject_hip_failure(0)
        trials.append({"victim": victim, "nth": nth, "compute_h_over_ranks": True, "failed": failed, "seconds": round(dt, 3), "msg": msg[:160]})
        if not failed:
            assert B.proof_write(got["raw"]) == want
        try:
            g.pk_free(spk)
        except B.MiError:
            pass
        g.close()
        g, spk = fresh()
        g.set_sharded_compute_h(True)
        got, _ = g.prove(spk, W, a, b, None, r, s, mode=1)
        assert B.proof_write(got["raw"]) == want, "the group must prove correctly (computeH over the ranks) after a failed call"
        g.set_sharded_compute_h(False)
    # a rank that passes a wrong witness length: refused on every rank, and the SAME group stays usable
    bad_failed = False
    t0 = time.time()
    try:
        g.prove(spk, W[:-1] if rank == 1 else W, a if lead else None, b if lead else None, cc if lead else None, r, s, mode=0)
    except B.MiError:
        bad_failed = True
    trials.append({"bad_argument_on_rank_1": True, "failed": bad_failed, "seconds": round(time.time() - t0, 3)})
    got, _ = g.prove(spk, W, a if lead else None, b if lead else None, cc if lead else None, r, s, mode=0)
    assert B.proof_write(got["raw"]) == want
    # the sharded MSM: a failure while one rank enqueues its share
    n = 20000
    pts = cref.gen_g1(n, 41); sc = cref.gen_scalars(n, 42, 1)
    lo, hi = B.shard_range(n, world, rank)
    c = g.ctx(0)
    dp, ds = c.to_dev(pts[lo:hi]), c.to_dev(sc[lo:hi])
    if rank == 1:
        lib.mi_debug_inject_hip_failure(2)
    msm_failed = False
    t0 = time.time()
    try:
        g.msm_dev([dp.ptr], [ds.ptr], [hi - lo], n, mode=1)
    except B.MiError:
        msm_failed = True
    lib.mi_debug_inject_hip_failure(0)
    trials.append({"msm_inject_on_rank_1": True, "failed": msm_failed, "seconds": round(time.time() - t0, 3)})
    assert np.array_equal(g.msm_dev([dp.ptr], [ds.ptr], [hi - lo], n, mode=1), cref.msm_g1(pts, sc)), "the group must still work after the refused MSM"
    g.pk_free(spk)
    g.close()
    return {"ok": True, "trials": trials}


def scenario_dead_peer(B, rank, world, seed_hex):
    """rank 1 ends before the collective call: rank 0 must get an error at its deadline (MI_GROUP_TIMEOUT_MS), not hang"""
    pk, W, a, b, cc, r, s, want = workload(12, 9300)
    g = B.Group.rank(0, rank, world, uid_of(seed_hex, 0), transport=3)
    g.exchange_selftest(4096)
    spk = g.pk_load(pk)
    if rank != 0:
        sys.stdout.write(json.dumps({"ok": True, "left": True}) + "\n")
        sys.stdout.flush()
        os._exit(0)   # no clean-up on purpose
    time.sleep(0.5)
    t0 = time.time()
    try:
        g.prove(spk, W, a, b, cc, r, s, mode=0)
        return {"ok": False, "error": "the prove succeeded without its peer"}
    except B.MiError as e:
        dt = time.time() - t0
        broken = False
        try:
            g.exchange_selftest(4096)
        except B.MiError as e2:
            broken = "destroy the group" in str(e2)
        g.close()
        return {"ok": True, "seconds": round(dt, 3), "msg": str(e)[:200], "group_refuses_later_calls": broken}


def scenario_stale_segment(B, rank, world, seed_hex):
    """the test has left a segment of an EARLIER group under this id's name (magic, shape and a full `attached` count in place, no creator
    alive): the ranks must find their own segment all the same"""
    g = B.Group.rank(0, rank, world, uid_of(seed_hex, 0), transport=3)
    try:
        g.exchange_selftest(4096)
        g.exchange_selftest((1 << 20) + 3)
    finally:
        g.close()
    return {"ok": True}


def main():
    scenario, seed_hex = sys.argv[1], sys.argv[2]
    rank, world = int(os.environ["MI_RANK"]), int(os.environ["MI_WORLD"])
    B = load_binding()
    try:
        res = {"parity": scenario_parity, "inject": scenario_inject, "dead_peer": scenario_dead_peer, "stale_segment": scenario_stale_segment}[scenario](B, rank, world, seed_hex)
    except BaseException as e:
        import traceback
        res = {"ok": False, "error": f"{type(e).__name__}: {e}", "trace": traceback.format_exc()[-1500:]}
    res["rank"] = rank
    print(json.dumps(res), flush=True)
    sys.exit(0 if res.get("ok") else 1)


if __name__ == "__main__":
    main()
