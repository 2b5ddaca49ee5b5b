"""GPU parity (through the C-ABI) of the device-group entry points: one MSM and one whole proof point-sharded over several
ranks (BASELINE configs[4]).  This box has ONE GPU, so the multi-rank cases name device 0 two or three times: same
partitioning, same per-rank Pippenger, same bucket exchange and slice sums, moved by same-process copies instead of RCCL
(csrc/group.hip picks the transport).  The RCCL calls themselves run in the world-1 group (grouped ncclSend / ncclRecv to
self).  Every result is compared byte for byte with the oracle and with the unsharded entry points."""
import os
import numpy as np
import pytest
import pyref as P
import cref
from helpers import *
from gpu_common import load_binding

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def B():
    return load_binding()


@pytest.fixture(scope="module", params=[(0,), "per-rank", (0, 0), (0, 0, 0)], ids=["world1-rccl", "world1-rccl-per-rank", "world2-samedev", "world3-samedev"])
def grp(B, request):
    # "per-rank": mi_group_create_rank with world 1 -- the NON-BLOCKING communicator of the one-rank-per-process flow: joining, every
    # grouped send / receive and every all-gather polled against MI_GROUP_TIMEOUT_MS (csrc/group.hip), which is what an 8-GPU node runs
    g = B.Group.rank(0, 0, 1, B.Group.unique_id(), transport=1) if request.param == "per-rank" else B.Group(list(request.param))
    yield g
    g.close()


def test_transport_selftest(grp):
    assert grp.world == grp.n_local
    assert grp.transport() == ("rccl" if grp.world == 1 else "peer-copy")
    grp.exchange_selftest(4096)
    grp.exchange_selftest(1 << 20)


@pytest.mark.parametrize("n,dist", [(5000, 1), (70001, 0), (3, 0)])
@pytest.mark.parametrize("mode", [0, 1])
def test_sharded_msm_g1_vs_oracle(grp, n, dist, mode):
    pts = cref.gen_g1(n, 900 + n); sc = cref.gen_scalars(n, 901 + n, dist)
    if n > 100:
        pts[3] = 0; sc[1] = fr_arr([P.R_MOD - 1])[0]; sc[2] = 0; pts[6] = pts[5]; sc[6] = sc[5]
    assert np.array_equal(grp.msm_g1(pts, sc, mode=mode), cref.msm_g1(pts, sc))


@pytest.mark.parametrize("mode", [0, 1])
def test_sharded_msm_dev_g1_g2_vs_oracle(B, grp, mode):
    """pairs resident per rank (the entry point bench.py drives with one rank per process)"""
    n, n2 = 40000, 2500
    pts = cref.gen_g1(n, 31); sc = cref.gen_scalars(n, 32, 1)
    p2 = cref.gen_g2(n2, 33); s2 = cref.gen_scalars(n2, 34, 0)
    keep, pp, ss, nn, pp2, ss2, nn2 = [], [], [], [], [], [], []
    for r in range(grp.world):
        c = grp.ctx(r)
        lo, hi = B.shard_range(n, grp.world, r); lo2, hi2 = B.shard_range(n2, grp.world, r)
        for arr, dst in ((pts[lo:hi], pp), (sc[lo:hi], ss), (p2[lo2:hi2], pp2), (s2[lo2:hi2], ss2)):
            d = c.to_dev(arr); keep.append(d); dst.append(d.ptr)
        nn.append(hi - lo); nn2.append(hi2 - lo2)
    assert np.array_equal(grp.msm_dev(pp, ss, nn, n, mode=mode), cref.msm_g1(pts, sc))
    assert np.array_equal(grp.msm_dev(pp2, ss2, nn2, n2, mode=mode, g2=True), cref.msm_g2(p2, s2))
    for d in keep:
        d.free()


@pytest.mark.parametrize("mode", [0, 1])
def test_sharded_msm_g2_host_arrays_vs_oracle(grp, mode):
    """mi_msm_g2_sharded: the G2 twin of the host-array entry point"""
    n = 3001
    pts = cref.gen_g2(n, 77); sc = cref.gen_scalars(n, 78, 1)
    pts[4] = 0; sc[2] = 0
    assert np.array_equal(grp.msm_g2(pts, sc, mode=mode), cref.msm_g2(pts, sc))


def _slices(B, grp, pk, keep):
    """per local rank: device copies of that rank's slices of the five point arrays (what mi_pk_load_sharded_dev adopts)"""
    N = 1 << pk["log_n"]
    ia, ib = np.asarray(pk["infinity_a"]), np.asarray(pk["infinity_b"])
    cw = set(int(x) for x in (pk.get("committed_wires") if pk.get("committed_wires") is not None else []))
    in_k = np.array([j >= pk["nb_public"] and j not in cw for j in range(pk["nb_wires"])])
    ca, cb, ck = np.concatenate([[0], np.cumsum(ia == 0)]), np.concatenate([[0], np.cumsum(ib == 0)]), np.concatenate([[0], np.cumsum(in_k)])
    out = []
    for r in range(grp.world):
        c = grp.ctx(r)
        lo, hi = grp.wire_range(pk["nb_wires"], r); zlo, zhi = B.shard_range(N - 1, grp.world, r)   # wires by the group's lead share, Z evenly
        sl = {}
        for name, arr in (("g1_a", pk["g1_a"][ca[lo]:ca[hi]]), ("g1_b", pk["g1_b"][cb[lo]:cb[hi]]), ("g1_k", pk["g1_k"][ck[lo]:ck[hi]]),
                          ("g1_z", pk["g1_z"][zlo:zhi]), ("g2_b", pk["g2_b"][cb[lo]:cb[hi]])):
            d = c.to_dev(arr); keep.append(d); sl[name] = (d.ptr, arr.shape[0])
        out.append(sl)
    return out


@pytest.mark.parametrize("log_n,nb_public,n_committed", [(13, 5, 0), (15, 4097, 23)])
@pytest.mark.parametrize("mode", [0, 1])
def test_sharded_prove_device_slices_and_device_inputs(B, grp, log_n, nb_public, n_committed, mode):
    """mi_pk_load_sharded_dev (every rank's slices already on its device) + mi_groth16_prove_sharded_dev (W ranges, a, b, c in HBM):
    the shape bench.py times for BASELINE configs[4]; bytes == oracle"""
    N = 1 << log_n
    nb_wires, n_constraints = N - 50, N - 10
    pk = synthetic_pk(log_n, nb_wires, nb_public, 8000 + log_n, n_committed=n_committed)
    W = cref.gen_scalars(nb_wires, 11, 1)
    a = cref.gen_scalars(n_constraints, 12, 1); b = cref.gen_scalars(n_constraints, 13, 0); c = cref.field_op(0, 2, a, b)
    r, s = cref.gen_scalars(2, 14, 0)
    want = cref.proof_write(cref.prove(pk, W, a, b, c, r, s)["raw"])
    keep = []
    spk = grp.pk_load_dev(pk, _slices(B, grp, pk, keep))
    Wp = []
    for rk in range(grp.world):
        lo, hi = grp.wire_range(nb_wires, rk)
        d = grp.ctx(rk).to_dev(W[lo:hi]); keep.append(d); Wp.append(d.ptr)
    c0 = grp.ctx(0)
    da, db, dc = c0.to_dev(a), c0.to_dev(b), c0.to_dev(c); keep += [da, db, dc]
    got, st = grp.prove_dev(spk, Wp, nb_wires, da.ptr, db.ptr, dc.ptr, n_constraints, r, s, mode=mode)
    got2, _ = grp.prove(spk, W, a, b, c, r, s, mode=mode)   # the same sharded key through the host-input entry point
    grp.pk_free(spk)
    for d in keep:
        d.free()
    assert B.proof_write(got["raw"]) == want and B.proof_write(got2["raw"]) == want
    bad = dict(pk); bad["nb_public"] = nb_public + 1   # slice counts no longer match the masks / public set
    keep2 = []
    with pytest.raises(B.MiError):
        grp.pk_load_dev(bad, _slices(B, grp, pk, keep2))
    for d in keep2:
        d.free()


def test_group_refuses_overlapping_calls(B, grp):
    """a group serves one call at a time: a call that arrives while another is running gets MI_EINVAL and disturbs nothing"""
    import threading
    n = 1 << 21
    c0 = grp.ctx(0)
    dp, ds = c0.gen_g1(n, 5150), c0.gen_scalars(n, 5151, 0)
    pts, sc = dp.download((n, 8)), ds.download((n, 4)); dp.free(); ds.free()
    want = grp.msm_g1(pts, sc)   # also sizes the workspaces
    res = {}
    th = threading.Thread(target=lambda: res.update(got=grp.msm_g1(pts, sc)))
    th.start()
    refused = ok = 0
    while th.is_alive():
        rc = grp.lib.mi_group_exchange_selftest(grp.h, 4096)
        assert rc in (0, -1)
        refused += rc == -1; ok += rc == 0
    th.join()
    assert np.array_equal(res["got"], want)
    assert refused >= 1, (refused, ok)
    grp.exchange_selftest(4096)   # and the group still works


def _toy():
    z = np.load(os.path.join(GOLD, "prove_toy1000.npz"))
    pk = {k: z[k] for k in ("g1_a", "g1_b", "g1_k", "g1_z", "g2_b", "alpha1", "beta1", "delta1", "beta2", "delta2", "infinity_a", "infinity_b")}
    pk.update(log_n=int(z["log_n"]), nb_public=int(z["nb_public"]), nb_wires=int(z["nb_wires"]))
    return z, pk


@pytest.mark.parametrize("mode", [0, 1])
def test_sharded_prove_golden_toy(B, grp, mode):
    """the committed 1000-constraint golden proof (trapdoor-checked by the oracle), proved over the group's ranks"""
    z, pk = _toy()
    spk = grp.pk_load(pk)
    proof, _ = grp.prove(spk, z["W"], z["a"], z["b"], z["c"], z["r"], z["s"], mode=mode)
    grp.pk_free(spk)
    assert B.proof_write(proof["raw"]) == bytes(z["proof_bytes"])


@pytest.mark.parametrize("log_n,nb_public,n_committed,knob", [(12, 5, 0, (0, 0, 0)), (14, 4097, 37, (0, 0, 0)), (16, 300, 0, (0, 0, 0)),
                                                             (15, 100, 11, (17, 18, 17))])
@pytest.mark.parametrize("mode", [0, 1])
def test_sharded_prove_vs_oracle_and_unsharded(B, grp, log_n, nb_public, n_committed, knob, mode):
    """synthetic keys of the bench's shape (infinity masks, public wires, committed wires removed from K), generic plan and
    forced fixed-base tables: sharded proof bytes == oracle == the unsharded mi_groth16_prove"""
    N = 1 << log_n
    nb_wires, n_constraints = N - 50, N - 10
    pk = synthetic_pk(log_n, nb_wires, nb_public, 7000 + log_n, n_committed=n_committed)
    W = cref.gen_scalars(nb_wires, 1, 1)
    a = cref.gen_scalars(n_constraints, 2, 1); b = cref.gen_scalars(n_constraints, 3, 0); c = cref.field_op(0, 2, a, b)
    r, s = cref.gen_scalars(2, 4, 0)
    want = cref.proof_write(cref.prove(pk, W, a, b, c, r, s)["raw"])
    for i in range(grp.n_local):
        assert grp.lib.mi_debug_set_prove_fixed_base(grp.ctx(i).h, *knob) == 0
    try:
        spk = grp.pk_load(pk)
    finally:
        for i in range(grp.n_local):
            assert grp.lib.mi_debug_set_prove_fixed_base(grp.ctx(i).h, 0, 0, 0) == 0
    got, st = grp.prove(spk, W, a, b, c, r, s, mode=mode)
    grp.pk_free(spk)
    assert B.proof_write(got["raw"]) == want
    c0 = grp.ctx(0)
    pkh = c0.pk_load(pk)
    solo, _ = c0.prove(pkh, W, a, b, c, r, s)
    c0.pk_free(pkh)
    assert B.proof_write(solo["raw"]) == want
    assert st["total_ms"] > 0


def test_group_argument_errors(B, grp):
    z, pk = _toy()
    spk = grp.pk_load(pk)
    with pytest.raises(B.MiError):   # witness of the wrong length
        grp.prove(spk, z["W"][:-1], z["a"], z["b"], z["c"], z["r"], z["s"])
    with pytest.raises(B.MiError):   # unknown mode
        grp.prove(spk, z["W"], z["a"], z["b"], z["c"], z["r"], z["s"], mode=2)
    grp.pk_free(spk)
    bad = dict(pk); bad["g1_a"] = pk["g1_a"][:-1]
    with pytest.raises(B.MiError):   # point counts that do not match the masks
        grp.pk_load(bad)
    if grp.world > 1:
        # an MSM with an EMPTY rank (one pair, several ranks): rounds 1-4 refused it in mode 1; an empty deferred MSM now leaves an
        # all-infinity bucket array and the exchange runs like any other
        pts = cref.gen_g1(1, 5); sc = cref.gen_scalars(1, 6, 0)
        for mode in (1, 0):
            assert np.array_equal(grp.msm_g1(pts, sc, mode=mode), cref.msm_g1(pts, sc))


def test_sharded_prove_at_2p22_two_ranks_automatic_tables(B):
    """N = 2^22: every part has >= 2^20 points per MSM, so mi_pk_load_sharded builds the fixed-base tables by itself (c = 19 / 18 /
    20, as at the benchmark size).  2 ranks on device 0, both modes, against the oracle and the unsharded proof of the same inputs"""
    g = B.Group([0, 0])
    try:
        log_n = 22
        N = 1 << log_n
        nb_wires, nb_public, n_constraints = N - 1000, 4097, N - 100
        c0 = g.ctx(0)
        rng = np.random.default_rng(5)
        inf_a = (rng.integers(0, 100, nb_wires) < 10).astype(np.uint8); inf_b = (rng.integers(0, 100, nb_wires) < 50).astype(np.uint8)
        na, nb, nk = int((inf_a == 0).sum()), int((inf_b == 0).sum()), nb_wires - nb_public

        def pull(d, shape):
            out = d.download(shape); d.free(); return out
        small = cref.gen_g1(3, 6); small2 = cref.gen_g2(2, 7)
        pk = {"log_n": log_n, "nb_public": nb_public, "nb_wires": nb_wires, "g1_a": pull(c0.gen_g1(na, 1), (na, 8)), "g1_b": pull(c0.gen_g1(nb, 2), (nb, 8)),
              "g1_k": pull(c0.gen_g1(nk, 3), (nk, 8)), "g1_z": pull(c0.gen_g1(N, 4), (N, 8)), "g2_b": pull(c0.gen_g2(nb, 5), (nb, 16)),
              "alpha1": small[0], "beta1": small[1], "delta1": small[2], "beta2": small2[0], "delta2": small2[1], "infinity_a": inf_a, "infinity_b": inf_b}
        W = pull(c0.gen_scalars(nb_wires, 8, 1), (nb_wires, 4)); a = pull(c0.gen_scalars(n_constraints, 9, 1), (n_constraints, 4))
        b = pull(c0.gen_scalars(n_constraints, 10, 0), (n_constraints, 4)); c = cref.field_op(0, 2, a, b)
        r, s = cref.gen_scalars(2, 11, 0)
        pkh = c0.pk_load(pk)
        solo, _ = c0.prove(pkh, W, a, b, c, r, s)
        c0.pk_free(pkh)
        want = B.proof_write(solo["raw"])
        assert want == cref.proof_write(cref.prove(pk, W, a, b, c, r, s)["raw"])
        spk = g.pk_load(pk)
        for mode in (0, 1):
            got, st = g.prove(spk, W, a, b, c, r, s, mode=mode)
            assert B.proof_write(got["raw"]) == want, mode
            print(f"sharded prove N=2^{log_n}, 2 ranks on one device, mode {mode}: {st['total_ms']:.1f} ms")
        g.pk_free(spk)
    finally:
        g.close()


def test_rccl_per_rank_group_injected_failures_break_or_spare_the_group_but_never_hang(B):
    """the per-rank RCCL group (non-blocking communicator, world 1): the n-th checked call of a sharded MSM / prove fails for n = 1, 2, ...
    Every call returns within seconds; a failure outside an exchange leaves the group usable, a failure inside one (ncclSend, the
    all-gather, the exchange stream) aborts the communicator and every later call is refused until the group is recreated; the
    recreated group gives the oracle's bytes."""
    import time
    n = 30000
    pts = cref.gen_g1(n, 51); sc = cref.gen_scalars(n, 52, 1)
    want = cref.msm_g1(pts, sc)
    fresh = lambda: B.Group.rank(0, 0, 1, B.Group.unique_id(), transport=1)
    g = fresh()
    c = g.ctx(0)
    dp, ds = c.to_dev(pts), c.to_dev(sc)
    outcomes = set()
    try:
        assert np.array_equal(g.msm_dev([dp.ptr], [ds.ptr], [n], n, mode=1), want)
        for nth in list(range(1, 40)) + [60, 90, 140]:
            assert g.lib.mi_debug_inject_hip_failure(nth) == 0
            t0 = time.time()
            failed = False
            try:
                got = g.msm_dev([dp.ptr], [ds.ptr], [n], n, mode=nth & 1)
            except B.MiError:
                failed = True
            g.lib.mi_debug_inject_hip_failure(0)
            assert time.time() - t0 < 20, f"call {nth} took {time.time() - t0:.1f} s"
            if not failed:
                assert np.array_equal(got, want), nth
                outcomes.add("beyond")
                continue
            try:   # usable or broken?
                again = g.msm_dev([dp.ptr], [ds.ptr], [n], n, mode=1)
                assert np.array_equal(again, want), nth
                outcomes.add("spared")
            except B.MiError as e:
                assert "destroy the group" in str(e), str(e)
                outcomes.add("broken")
                dp.free(); ds.free(); g.close()
                g = fresh(); c = g.ctx(0)
                dp, ds = c.to_dev(pts), c.to_dev(sc)
                assert np.array_equal(g.msm_dev([dp.ptr], [ds.ptr], [n], n, mode=1), want), nth
        assert "broken" in outcomes and "beyond" in outcomes, outcomes   # the sweep reached the exchange's calls and ran past the last checked call
    finally:
        g.lib.mi_debug_inject_hip_failure(0)
        dp.free(); ds.free(); g.close()


@pytest.mark.parametrize("share", [0, 500, 1000, 0xFFFFFFFF])
@pytest.mark.parametrize("mode", [0, 1])
def test_lead_share_of_the_wires_gives_the_same_proof(B, grp, share, mode):
    """mi_group_set_lead_share: rank 0 (which also runs computeH) takes 0, half, all of an even wire share, or the automatic one; host
    arrays and device slices, both modes (share 0: the lead's wire MSMs are EMPTY and still take part in the bucket exchange); bytes ==
    oracle; the cut is what mi_group_wire_range says"""
    log_n = 13
    N = 1 << log_n
    nb_wires, n_constraints = N - 50, N - 10
    pk = synthetic_pk(log_n, nb_wires, 300, 8700, n_committed=11)
    W = cref.gen_scalars(nb_wires, 21, 1)
    a = cref.gen_scalars(n_constraints, 22, 1); b = cref.gen_scalars(n_constraints, 23, 0); c = cref.field_op(0, 2, a, b)
    r, s = cref.gen_scalars(2, 24, 0)
    want = cref.proof_write(cref.prove(pk, W, a, b, c, r, s)["raw"])
    try:
        grp.set_lead_share(share)
        spans = [grp.wire_range(nb_wires, rk) for rk in range(grp.world)]
        assert spans[0][0] == 0 and spans[-1][1] == nb_wires and all(spans[i][1] == spans[i + 1][0] for i in range(grp.world - 1))
        if grp.world > 1 and share <= 1000:
            assert spans[0][1] == nb_wires * share // (1000 * grp.world)
        if share == 0xFFFFFFFF:
            assert spans[0][1] == {1: nb_wires, 2: nb_wires * 500 // 2000}.get(grp.world, 0)
        spk = grp.pk_load(pk)
        got, _ = grp.prove(spk, W, a, b, c, r, s, mode=mode)
        grp.pk_free(spk)
        assert B.proof_write(got["raw"]) == want, (share, mode, "host arrays")
        keep = []
        spk = grp.pk_load_dev(pk, _slices(B, grp, pk, keep))
        Wp = []
        for rk in range(grp.world):
            d = grp.ctx(rk).to_dev(W[spans[rk][0]:spans[rk][1]]); keep.append(d); Wp.append(d.ptr)
        c0 = grp.ctx(0)
        da, db = c0.to_dev(a), c0.to_dev(b); keep += [da, db]
        got, _ = grp.prove_dev(spk, Wp, nb_wires, da.ptr, db.ptr, None, n_constraints, r, s, mode=mode)
        grp.pk_free(spk)
        for d in keep:
            d.free()
        assert B.proof_write(got["raw"]) == want, (share, mode, "device slices")
    finally:
        grp.set_lead_share(0xFFFFFFFF)


def test_a_key_remembers_the_wire_cut_it_was_loaded_with(B):
    """a prove under another lead share than the key's is refused -- mi_group_wire_range answers with the group's CURRENT share,
    so a caller of the _dev entry points would cut W unlike the key's parts -- and accepted again once the share is the key's"""
    log_n = 12
    N = 1 << log_n
    nb_wires, n_constraints = N - 50, N - 10
    pk = synthetic_pk(log_n, nb_wires, 100, 8900)
    W = cref.gen_scalars(nb_wires, 41, 1); a = cref.gen_scalars(n_constraints, 42, 1); b = cref.gen_scalars(n_constraints, 43, 0); c = cref.field_op(0, 2, a, b)
    r, s = cref.gen_scalars(2, 44, 0)
    want = cref.proof_write(cref.prove(pk, W, a, b, c, r, s)["raw"])
    g = B.Group([0] * 4)
    try:
        assert g.wire_range(nb_wires, 0) == (0, 0)                       # automatic, 4 ranks: no wires on the lead
        spk = g.pk_load(pk)
        assert B.proof_write(g.prove(spk, W, a, b, c, r, s, mode=1)[0]["raw"]) == want
        g.set_sharded_compute_h(True)                                     # computeH over the ranks on the same key: the cut does not move
        assert B.proof_write(g.prove(spk, W, a, b, c, r, s, mode=1)[0]["raw"]) == want
        g.set_sharded_compute_h(False)
        g.set_lead_share(1000)                                            # not this key's cut any more
        assert g.wire_range(nb_wires, 0) == (0, nb_wires // 4)
        for mode in (0, 1):
            with pytest.raises(B.MiError, match="reload the key"):
                g.prove(spk, W, a, b, c, r, s, mode=mode)
        g.set_lead_share(0)                                               # the key's own cut, stated explicitly: accepted
        assert B.proof_write(g.prove(spk, W, a, b, c, r, s, mode=0)[0]["raw"]) == want
        g.pk_free(spk)
    finally:
        g.close()


def test_sharded_msm_with_ranks_that_hold_no_pairs(B):
    """fewer pairs than ranks: some ranks' shares are EMPTY; mode 1 exchanges their (all-infinity) bucket arrays like any other"""
    g = B.Group([0, 0, 0])
    try:
        for n in (1, 2):
            pts = cref.gen_g1(n, 61); sc = cref.gen_scalars(n, 62, 0)
            for mode in (0, 1):
                assert np.array_equal(g.msm_g1(pts, sc, mode=mode), cref.msm_g1(pts, sc)), (n, mode)
    finally:
        g.close()


@pytest.mark.parametrize("world", [2, 4, 8])
@pytest.mark.parametrize("log_n,n_constraints", [(6, 64), (10, 1000), (13, 8192 - 77), (16, 40000)])
def test_compute_h_over_the_ranks_equals_the_oracle(B, world, log_n, n_constraints):
    """mi_compute_h_sharded_dev: six transforms as local size-N/W transforms + cross-rank steps between all-to-alls (csrc/ntt_cross.hip):
    every coefficient of h against the oracle's computeH, with c given and with c = a o b formed on the devices; ragged n_constraints
    (whole slices of zero padding), 2 / 4 / 8 ranks on device 0 (same-process copies)"""
    N = 1 << log_n
    M = N // world
    a = cref.gen_scalars(n_constraints, 7700 + log_n, 1); b = cref.gen_scalars(n_constraints, 7701 + log_n, 0)
    c_good = cref.field_op(0, 2, a, b)
    c_free = cref.gen_scalars(n_constraints, 7702 + log_n, 0)   # computeH does not assume a o b = c
    g = B.Group([0] * world)
    try:
        for c_host in (c_free, None, c_good):
            want = cref.compute_h(log_n, a, b, c_good if c_host is None else c_host)
            keep, ap, bp, cp, hp, hd = [], [], [], [], [], []
            for r in range(world):
                ctx = g.ctx(r)
                lo, hi = min(r * M, n_constraints), min((r + 1) * M, n_constraints)
                for arr, dst in ((a, ap), (b, bp), (c_host, cp)):
                    if arr is None:
                        continue
                    d = ctx.to_dev(arr[lo:hi]) if hi > lo else ctx.alloc(32); keep.append(d); dst.append(d.ptr)
                h = ctx.alloc(32 * M); keep.append(h); hp.append(h.ptr); hd.append(h)
            g.compute_h_sharded_dev(log_n, ap, bp, cp if c_host is not None else None, n_constraints, hp)
            got = np.concatenate([h.download((M, 4)) for h in hd])
            for d in keep:
                d.free()
            assert np.array_equal(got, want), (world, log_n, n_constraints, c_host is None)
    finally:
        g.close()


def test_compute_h_over_the_ranks_refuses_what_it_cannot_cut(B):
    for devs, log_n in (([0, 0, 0], 10), ([0, 0, 0, 0], 3), ([0], 10)):
        g = B.Group(devs)
        try:
            ctx = g.ctx(0)
            d = ctx.alloc(32 << log_n)
            with pytest.raises(B.MiError):
                g.compute_h_sharded_dev(log_n, [d.ptr] * len(devs), [d.ptr] * len(devs), None, 1 << log_n, [d.ptr] * len(devs))
            d.free()
        finally:
            g.close()


@pytest.mark.parametrize("world,mode", [(2, 0), (2, 1), (4, 0), (4, 1), (8, 1)])
def test_sharded_prove_with_compute_h_over_the_ranks(B, world, mode):
    """mi_group_set_sharded_compute_h + mi_groth16_prove_sharded (host arrays) and mi_groth16_prove_sharded_slices_dev (row slices of a, b,
    c per rank): computeH runs over all ranks and every rank's Z MSM reads the h slice born on it; bytes == oracle, with c given and
    with c formed on the devices; the lead's wire share 0 and even"""
    log_n = 13
    N = 1 << log_n
    M = N // world
    nb_wires, n_constraints = N - 50, N - 1000
    pk = synthetic_pk(log_n, nb_wires, 300, 8800 + world, n_committed=7)
    W = cref.gen_scalars(nb_wires, 31, 1)
    a = cref.gen_scalars(n_constraints, 32, 1); b = cref.gen_scalars(n_constraints, 33, 0); c = cref.field_op(0, 2, a, b)
    r, s = cref.gen_scalars(2, 34, 0)
    want = cref.proof_write(cref.prove(pk, W, a, b, c, r, s)["raw"])
    g = B.Group([0] * world)
    try:
        for share in (0, 1000):
            g.set_lead_share(share)
            g.set_sharded_compute_h(True)
            spk = g.pk_load(pk)
            for cc in (c, None):
                got, st = g.prove(spk, W, a, b, cc, r, s, mode=mode)
                assert B.proof_write(got["raw"]) == want, (world, mode, share, cc is None, "host arrays")
            g.set_sharded_compute_h(False)
            got, _ = g.prove(spk, W, a, b, c, r, s, mode=mode)   # and the lead's computeH again on the same key
            assert B.proof_write(got["raw"]) == want
            g.pk_free(spk)
            keep = []
            spk = g.pk_load_dev(pk, _slices(B, g, pk, keep))
            Wp, ap, bp, cp = [], [], [], []
            for rk in range(world):
                ctx = g.ctx(rk)
                wlo, whi = g.wire_range(nb_wires, rk)
                lo, hi = min(rk * M, n_constraints), min((rk + 1) * M, n_constraints)
                for arr, dst, sl in ((W, Wp, slice(wlo, whi)), (a, ap, slice(lo, hi)), (b, bp, slice(lo, hi)), (c, cp, slice(lo, hi))):
                    d = ctx.to_dev(arr[sl]) if sl.stop > sl.start else ctx.alloc(32); keep.append(d); dst.append(d.ptr)
            for cps in (cp, None):
                got, _ = g.prove_slices_dev(spk, Wp, nb_wires, ap, bp, cps, n_constraints, r, s, mode=mode)
                assert B.proof_write(got["raw"]) == want, (world, mode, share, cps is None, "row slices")
            g.pk_free(spk)
            for d in keep:
                d.free()
    finally:
        g.close()
