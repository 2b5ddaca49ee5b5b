"""GPU: the ONE-RANK-PER-PROCESS device group with world = 2 on one GPU (BASELINE configs[4]) -- the shape an 8-GPU node
runs under torch.distributed.run, rehearsed here with two processes on device 0 over the host-staged transport
(mi_group_create_rank_ex(..., MI_GROUP_TRANSPORT_HOST)): RCCL refuses two ranks on one device, the rest of csrc/group.hip's
multi-process flow (rank0 != 0, lead / non-lead, agreements, all-gathers, batches with remote ranks) is the same code.
The rank processes are tests/group_rank_worker.py, started by tests/rank_launcher.py (a process that never touches the GPU; see
tests/conftest.py); each rank checks its own results against the oracle."""
import json
import os
import sys
import pytest
from gpu_common import ROOT

pytestmark = pytest.mark.gpu
WORKER = os.path.join(ROOT, "tests", "group_rank_worker.py")


def _run(rank_launcher, scenario, world=2, env=None, timeout=420, seed=None):
    seed = seed or os.urandom(16).hex()
    e = {"OMP_WAIT_POLICY": "passive", "OMP_NUM_THREADS": "8"}   # the oracle in every rank process: its OpenMP barriers under a CPU quota, see oracle/cref.py
    e.update(env or {})
    ranks = rank_launcher.run([sys.executable, WORKER, scenario, seed], world, env=e, timeout=timeout)
    res = []
    for i, r in enumerate(ranks):
        lines = [ln for ln in r["stdout"].splitlines() if ln.lstrip().startswith("{")]
        assert lines, f"rank {i} printed no result (rc {r['rc']}): {r['stdout'][-2000:]} {r['stderr'][-2000:]}"
        res.append((r["rc"], json.loads(lines[-1]), r))
    return res


def test_world2_two_processes_parity_vs_oracle(rank_launcher):
    """transport self-test, sharded G1 / G2 MSM in both modes, sharded prove at N = 2^16 in both modes (host arrays and device
    slices, c given and c formed on the device): every rank's bytes equal the oracle's"""
    res = _run(rank_launcher, "parity")
    for rc, j, raw in res:
        assert rc == 0 and j.get("ok"), f"{j} {raw['stderr'][-1500:]}"
        assert j["checks"] == ["selftest", "msm_g1_g2_both_modes", "prove_2p16_host_and_device_both_modes", "lead_share_0_500_1000_and_disagreement", "compute_h_over_the_ranks"]


def test_world3_three_processes_parity_vs_oracle(rank_launcher):
    """the same with three ranks (a middle rank that is neither the lead nor the last; uneven slices)"""
    res = _run(rank_launcher, "parity", world=3)
    for rc, j, raw in res:
        assert rc == 0 and j.get("ok"), f"{j} {raw['stderr'][-1500:]}"


def test_world4_four_processes_parity_vs_oracle(rank_launcher):
    """four ranks: computeH over the ranks with two cross-rank butterfly stages, the lead without wires (automatic share)"""
    res = _run(rank_launcher, "parity", world=4, timeout=600)
    for rc, j, raw in res:
        assert rc == 0 and j.get("ok"), f"{j} {raw['stderr'][-1500:]}"


def test_local_failure_on_one_rank_is_an_error_on_every_rank(rank_launcher):
    """mi_debug_inject_hip_failure on ONE rank of a 2-rank group: every rank returns non-zero, within seconds (no rank is left waiting in
    an exchange), and the recreated group proves the oracle's bytes again"""
    res = _run(rank_launcher, "inject", env={"MI_GROUP_TIMEOUT_MS": "20000"})
    for rc, j, raw in res:
        assert rc == 0 and j.get("ok"), f"{j} {raw['stderr'][-1500:]}"
    t0, t1 = res[0][1]["trials"], res[1][1]["trials"]
    assert len(t0) == len(t1)
    n_failed = 0
    for x, y in zip(t0, t1):
        assert x["failed"] == y["failed"], f"the ranks disagree on the outcome of a call: {x} / {y}"
        if x["failed"]:
            n_failed += 1
            assert x["seconds"] < 8 and y["seconds"] < 8, f"a rank waited for its deadline instead of being told: {x} / {y}"
    assert n_failed >= 10, f"the injected failures must have hit: {t0}"
    assert sum(1 for x in t0 if x.get("compute_h_over_ranks") and x["failed"]) >= 4, f"... also inside computeH over the ranks: {t0}"
    assert t0[-2]["failed"] and t0[-1]["failed"]   # the wrong witness length on rank 1; the MSM with a failing rank


def test_dead_peer_is_a_timeout_not_a_hang(rank_launcher):
    res = _run(rank_launcher, "dead_peer", env={"MI_GROUP_TIMEOUT_MS": "3000"}, timeout=180)
    rc0, j0, raw0 = res[0]
    assert rc0 == 0 and j0.get("ok"), f"{j0} {raw0['stderr'][-1500:]}"
    assert j0["seconds"] < 15 and "timeout" in j0["msg"] and j0["group_refuses_later_calls"]


def test_a_leftover_segment_under_the_same_id_is_not_mistaken_for_the_groups_own(rank_launcher):
    """a reused group id after a run that died before rank 0 unlinked its segment -- the leftover's magic and world match and
    its attached counter already reads W.  The ranks tell it from their own by the name's inode and the creator's heartbeat."""
    import hashlib
    import struct
    seed = os.urandom(16).hex()
    h = hashlib.sha512((seed + ":0").encode()).digest()
    uid = (h + h)[:128]
    f = 1469598103934665603
    for b in uid:
        f = ((f ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    path = "/dev/shm/mi355x_grp_%016x" % f
    world, nslot, chunk = 2, 4, 1 << 20
    with open(path, "wb") as fh:   # magic | world | nslot | (pad) | chunk | attached | poisoned | beat: ShmHeader of csrc/group.hip
        fh.write(struct.pack("<IIIIQIIQ", 0x6d693335, world, nslot, 0, chunk, world, 0, 12345))
        fh.truncate((1 << 20) + world * world * nslot * chunk)
    try:
        res = _run(rank_launcher, "stale_segment", env={"MI_GROUP_TIMEOUT_MS": "20000"}, timeout=120, seed=seed)
        for rc, j, raw in res:
            assert rc == 0 and j.get("ok"), f"{j} {raw['stderr'][-1500:]}"
    finally:
        if os.path.exists(path):
            os.unlink(path)
