"""GPU parity (through the C-ABI) of the prover pool mi_prover_*: several proofs in flight on one device must give, job
for job, the bytes the oracle gives for the same (pk, witness, r, s) -- whatever the interleaving on the GPU."""
import os
import sys
import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(__file__))
from gpu_common import load_binding          # noqa: E402
from helpers import synthetic_pk            # noqa: E402
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
import cref                                   # noqa: E402

pytestmark = pytest.mark.gpu


def _jobs(log_n, n_jobs, seed):
    n = 1 << log_n
    nb_wires, nb_public, n_constraints = n - 13, 41, n - 5
    pk = synthetic_pk(log_n, nb_wires, nb_public, seed, n_committed=5)
    jobs = []
    for j in range(n_jobs):   # two distinct witnesses, every job its own blinding
        w = j % 2
        W = cref.gen_scalars(nb_wires, 10 + w, 1)
        a = cref.gen_scalars(n_constraints, 20 + w, 1); b = cref.gen_scalars(n_constraints, 30 + w, 0); c = cref.field_op(0, 2, a, b)
        r, s = cref.gen_scalars(2, 100 + j, 0)
        jobs.append((W, a, b, c, r, s))
    return pk, jobs


@pytest.mark.parametrize("in_flight", [1, 3])
def test_pool_proofs_equal_oracle_host_buffers(in_flight):
    B = load_binding()
    pk, jobs = _jobs(12, 7, 4100)
    want = [cref.proof_write(cref.prove(pk, *j)["raw"]) for j in jobs]
    pool = B.Prover(0, in_flight)
    assert pool.in_flight == in_flight
    c0 = pool.ctx(0)
    pkh = c0.pk_load(pk)
    tickets = [pool.submit(pkh, *j) for j in jobs]          # all queued before the first wait
    got = {}
    for t in reversed(tickets):                              # collected out of order
        got[t] = pool.wait(t)
    for t, w in zip(tickets, want):
        proof, st = got[t]
        assert B.proof_write(proof["raw"]) == w
        assert st["total_ms"] > 0
    c0.pk_free(pkh)
    pool.close()


def test_pool_device_buffers_match_single_context():
    """device-resident inputs shared by all jobs; the same proofs from a plain context, bit for bit"""
    B = load_binding()
    pk, jobs = _jobs(14, 6, 4200)
    pool = B.Prover(0, 2)
    c0 = pool.ctx(0)
    pkh = c0.pk_load(pk)
    dev = []
    for W, a, b, c, _, _ in jobs[:2]:
        dev.append(tuple(c0.to_dev(x) for x in (W, a, b, c)))
    c0.sync()
    nw, nc = jobs[0][0].shape[0], jobs[0][1].shape[0]
    tickets = []
    for j, (_, _, _, _, r, s) in enumerate(jobs):
        W, a, b, c = dev[j % 2]
        tickets.append(pool.submit(pkh, W.ptr, a.ptr, b.ptr, c.ptr, r, s, device=True, n_wires=nw, n_constraints=nc))
    got = [pool.wait(t)[0]["raw"].copy() for t in tickets]
    single = B.Context(0)
    for j, (W, a, b, c, r, s) in enumerate(jobs):
        ref, _ = single.prove(pkh, W, a, b, c, r, s)        # the key is shared between contexts: read-only during prove
        assert np.array_equal(ref["raw"], got[j])
    assert B.proof_write(got[0]) == cref.proof_write(cref.prove(pk, *jobs[0])["raw"])
    single.close()
    for t in dev:
        for x in t:
            x.free()
    c0.pk_free(pkh)
    pool.close()


def test_pool_reports_job_errors_and_keeps_running():
    B = load_binding()
    pk, jobs = _jobs(10, 2, 4300)
    pool = B.Prover(0, 2)
    c0 = pool.ctx(0)
    pkh = c0.pk_load(pk)
    W, a, b, c, r, s = jobs[0]
    bad = pool.submit(pkh, W[:-1], a, b, c, r, s)           # witness shorter than the key's wire count
    good = pool.submit(pkh, W, a, b, c, r, s)
    with pytest.raises(B.MiError, match="witness size"):
        pool.wait(bad)
    proof, _ = pool.wait(good)
    assert B.proof_write(proof["raw"]) == cref.proof_write(cref.prove(pk, *jobs[0])["raw"])
    with pytest.raises(B.MiError):
        pool.ctx(2)
    with pytest.raises(B.MiError):
        B.Prover(0, 0)
    c0.pk_free(pkh)
    pool.close()


def test_pool_at_2p20_overlapping_witnesses_match_single_context():
    """N = 2^20 (fixed-base tables active, kernels of different proofs genuinely overlapping): nine jobs over three witnesses
    through a pool of three; every proof equals the one a plain context computes alone for the same (witness, r, s)"""
    B = load_binding()
    log_n = 20
    N = 1 << log_n
    nb_wires, nb_public, n_constraints = N - 1000, 4097, N - 100
    pool = B.Prover(0, 3)
    c0 = pool.ctx(0)
    rng = np.random.default_rng(20)
    inf_a = (rng.integers(0, 100, nb_wires) < 10).astype(np.uint8); inf_b = (rng.integers(0, 100, nb_wires) < 50).astype(np.uint8)
    na, nb, nk = int((inf_a == 0).sum()), int((inf_b == 0).sum()), nb_wires - nb_public
    g1a, g1b, g1k, g1z, g2b = c0.gen_g1(na, 1), c0.gen_g1(nb, 2), c0.gen_g1(nk, 3), c0.gen_g1(N, 4), c0.gen_g2(nb, 5)
    small = c0.gen_g1(3, 6).download((3, 8)); small2 = c0.gen_g2(2, 7).download((2, 16))
    pk = {"log_n": log_n, "nb_public": nb_public, "nb_wires": nb_wires, "g1_a": (g1a.ptr, na), "g1_b": (g1b.ptr, nb), "g1_k": (g1k.ptr, nk),
          "g1_z": (g1z.ptr, N), "g2_b": (g2b.ptr, nb), "alpha1": small[0], "beta1": small[1], "delta1": small[2], "beta2": small2[0],
          "delta2": small2[1], "infinity_a": inf_a, "infinity_b": inf_b}
    pkh = c0.pk_load(pk, device_points=True)
    wit = []
    for w in range(3):
        W = c0.gen_scalars(nb_wires, 100 + w, w % 2)
        a = c0.gen_scalars(n_constraints, 200 + w, 1); b = c0.gen_scalars(n_constraints, 300 + w, 0)
        c = c0.alloc(32 * n_constraints); c0.field_op_dev(0, 2, c.ptr, a.ptr, b.ptr, n_constraints)
        wit.append((W, a, b, c))
    rs = c0.gen_scalars(18, 400, 0).download((18, 4))
    c0.sync()
    tickets = [pool.submit(pkh, *(x.ptr for x in wit[j % 3]), rs[2 * j], rs[2 * j + 1], device=True, n_wires=nb_wires, n_constraints=n_constraints)
               for j in range(9)]
    got = [pool.wait(t)[0]["raw"].copy() for t in tickets]
    single = B.Context(0)
    for j in range(9):
        ref, _ = single.prove(pkh, *(x.ptr for x in wit[j % 3]), rs[2 * j], rs[2 * j + 1], device=True, n_wires=nb_wires, n_constraints=n_constraints)
        assert np.array_equal(ref["raw"], got[j]), j
    assert not np.array_equal(got[0], got[1]) and not np.array_equal(got[0], got[3])   # different witness / different blinding
    single.close()
    c0.pk_free(pkh)
    for d in (g1a, g1b, g1k, g1z, g2b) + tuple(x for w in wit for x in w):
        d.free()
    pool.close()


def test_pool_host_inputs_at_2p20_early_handover_and_ledger():
    """N = 2^20 with HOST inputs (the cgo path): the upload stage hands a job to a worker as soon as W has arrived and a, b, c follow
    behind an event (csrc/pool.hip) -- eight jobs over two witnesses through a pool of three, bursts included (all submitted at once:
    the first jobs are picked up while their a, b, c are still on the PCIe bus); every proof equals the one a plain context computes
    alone.  Also: mi_get_mem_ledger accounts for what the key and the contexts hold."""
    B = load_binding()
    log_n = 20
    N = 1 << log_n
    nb_wires, nb_public, n_constraints = N - 1000, 4097, N - 100
    pool = B.Prover(0, 3)
    c0 = pool.ctx(0)
    rng = np.random.default_rng(21)
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
        W = c0.gen_scalars(nb_wires, 500 + w, 1); a = c0.gen_scalars(n_constraints, 600 + w, 1); b = c0.gen_scalars(n_constraints, 700 + w, 0)
        c = c0.alloc(32 * n_constraints); c0.field_op_dev(0, 2, c.ptr, a.ptr, b.ptr, n_constraints)
        host = (W.download((nb_wires, 4)), a.download((n_constraints, 4)), b.download((n_constraints, 4)), c.download((n_constraints, 4)))
        wit.append(((W, a, b, c), host))
    rs = c0.gen_scalars(16, 800, 0).download((16, 4))
    c0.sync()
    single = B.Context(0)
    want = [single.prove(pkh, *(x.ptr for x in wit[j % 2][0]), rs[2 * j], rs[2 * j + 1], device=True, n_wires=nb_wires, n_constraints=n_constraints)[0]["raw"].copy()
            for j in range(8)]
    single.close()
    for burst in range(2):   # two bursts: the second finds the pool idle again
        tickets = [pool.submit(pkh, *wit[j % 2][1], rs[2 * j], rs[2 * j + 1]) for j in range(8)]
        for j, t in enumerate(tickets):
            proof, st = pool.wait(t)
            assert np.array_equal(proof["raw"], want[j]), (burst, j)
            assert st["h2d_ms"] > 0
    led = c0.mem_ledger(pkh)
    assert led["key_bases"] + led["key_tables"] > 0 and led["ctx_msm"] > 0 and led["ctx_ntt_vectors"] >= 2 * 32 * N / 1e9 and led["key_indices"] > 0
    assert pool.ctx(1).mem_ledger()["key_bases"] == 0 and pool.ctx(1).mem_ledger()["ctx_msm"] > 0
    for (dev, _) in wit:
        for x in dev:
            x.free()
    for d in (g1a, g1b, g1k, g1z, g2b):
        d.free()
    c0.pk_free(pkh)
    pool.close()


def test_c_formed_on_the_device_gives_the_same_proofs():
    """c == NULL on every prove entry point: c = a o b is formed on the device (mi_groth16_prove[_dev], mi_prover_submit[_dev]); the
    proof bytes are those of the oracle, which is handed c explicitly"""
    B = load_binding()
    pk, jobs = _jobs(13, 4, 4300)
    want = [cref.proof_write(cref.prove(pk, *j)["raw"]) for j in jobs]
    pool = B.Prover(0, 2)
    c0 = pool.ctx(0)
    pkh = c0.pk_load(pk)
    tickets = [pool.submit(pkh, W, a, b, None, r, s) for (W, a, b, c, r, s) in jobs]        # host inputs, no c
    for t, w in zip(tickets, want):
        assert B.proof_write(pool.wait(t)[0]["raw"]) == w
    W, a, b, c, r, s = jobs[0]
    dW, da, db = c0.to_dev(W), c0.to_dev(a), c0.to_dev(b)
    c0.sync()
    t = pool.submit(pkh, dW.ptr, da.ptr, db.ptr, None, r, s, device=True, n_wires=W.shape[0], n_constraints=a.shape[0])   # device inputs, no c
    assert B.proof_write(pool.wait(t)[0]["raw"]) == want[0]
    single = B.Context(0)
    assert B.proof_write(single.prove(pkh, W, a, b, None, r, s)[0]["raw"]) == want[0]                                       # one context, host inputs
    assert B.proof_write(single.prove(pkh, dW.ptr, da.ptr, db.ptr, None, r, s, device=True, n_wires=W.shape[0], n_constraints=a.shape[0])[0]["raw"]) == want[0]
    h_full = single.compute_h(13, a, b, c)
    assert np.array_equal(single.compute_h(13, a, b, None), h_full)
    # a, b, c that do NOT satisfy a o b = c: the general path must still be gnark's computeH (c is used, not re-derived)
    c_other = cref.gen_scalars(a.shape[0], 77, 0)
    assert np.array_equal(single.compute_h(13, a, b, c_other), cref.compute_h(13, a, b, c_other))
    single.close()
    for d in (dW, da, db):
        d.free()
    c0.pk_free(pkh)
    pool.close()


def test_host_job_that_fails_before_its_gate_waits_for_its_uploads():
    """a host job handed to an idle worker as soon as W is resident can fail BEFORE it reaches the a, b, c gate (here: a
    witness whose length does not match the key).  The job must not complete -- its waiter frees the caller's a, b, c -- while the
    uploader is still copying them.  The job's h2d_ms is written by the uploader after its last copy: a job that completed early would
    report 0 (and, under a sanitizer, a use after free)."""
    B = load_binding()
    pk, jobs = _jobs(16, 2, 4400)
    W, a, b, c, r, s = jobs[0]
    pool = B.Prover(0, 2)
    c0 = pool.ctx(0)
    pkh = c0.pk_load(pk)
    assert B.proof_write(pool.wait(pool.submit(pkh, W, a, b, c, r, s))[0]["raw"]) == cref.proof_write(cref.prove(pk, W, a, b, c, r, s)["raw"])
    for _ in range(3):   # idle workers: the early hand-over happens
        t = pool.submit(pkh, W[:-1], a, b, c, r, s)
        with pytest.raises(B.MiError, match="witness size"):
            pool.wait(t)
    # and the pool goes on proving
    assert B.proof_write(pool.wait(pool.submit(pkh, W, a, b, c, r, s))[0]["raw"]) == cref.proof_write(cref.prove(pk, W, a, b, c, r, s)["raw"])
    c0.pk_free(pkh)
    pool.close()


@pytest.mark.parametrize("n_commitments,n_values", [(1, 3000), (2, 40000), (1, 1 << 15)])
def test_bsb22_through_the_pool_matches_oracle(n_commitments, n_values):
    """the proof the WHIR circuit really produces, through the pool (mi_prover_commit inside the "solve", mi_prover_submit_bsb22 after it):
    commitments, the folded proof of knowledge and Proof.WriteTo's 164 + 32 n bytes equal the oracle's; several jobs in flight; c = NULL"""
    B = load_binding()
    log_n = 13
    n = 1 << log_n
    nb_wires, nb_public, n_constraints = n - 13, 41, n - 5
    pk = synthetic_pk(log_n, nb_wires, nb_public, 5100 + n_commitments, n_committed=50)
    W = cref.gen_scalars(nb_wires, 11, 1)
    a = cref.gen_scalars(n_constraints, 21, 1); b = cref.gen_scalars(n_constraints, 31, 0); c = cref.field_op(0, 2, a, b)
    keys = [(cref.gen_g1(n_values, 800 + i), cref.gen_g1(n_values, 900 + i)) for i in range(n_commitments)]
    vals = [cref.gen_scalars(n_values, 1000 + i, 1) for i in range(n_commitments)]
    challenge = cref.gen_scalars(1, 77, 0)[0]
    pool = B.Prover(0, 2)
    c0 = pool.ctx(0)
    pkh = c0.pk_load(pk)
    peds = [c0.pedersen_pk_load(bs, sg) for bs, sg in keys]
    want_cm = np.stack([cref.pedersen_msm(bs, v) for (bs, _), v in zip(keys, vals)])
    want_pok = cref.pedersen_fold(np.stack([cref.pedersen_msm(sg, v) for (_, sg), v in zip(keys, vals)]), challenge)
    jobs = []
    for j in range(4):
        r, s = cref.gen_scalars(2, 200 + j, 0)
        cms = np.stack([pool.commit(p, v) for p, v in zip(peds, vals)])      # synchronous, as inside the solve
        assert np.array_equal(cms, want_cm)
        jobs.append((r, s, pool.submit_bsb22(pkh, W, a, b, None if j % 2 else c, r, s, list(zip(peds, vals)), challenge)))
    for r, s, t in reversed(jobs):
        proof, _ = pool.wait(t)
        assert np.array_equal(proof["pok"], want_pok)
        want = cref.proof_write(cref.prove(pk, W, a, b, c, r, s)["raw"], want_cm, want_pok)
        assert len(want) == 164 + 32 * n_commitments
        assert B.proof_write(proof["raw"], want_cm, proof["pok"]) == want
    # a commitment with more values than its key has points: the job fails with the key's message, the pool goes on
    t = pool.submit_bsb22(pkh, W, a, b, c, *cref.gen_scalars(2, 1, 0), [(peds[0], cref.gen_scalars(n_values + 1, 5, 0))], challenge)
    with pytest.raises(B.MiError):
        pool.wait(t)
    r, s = cref.gen_scalars(2, 300, 0)
    proof, _ = pool.wait(pool.submit_bsb22(pkh, W, a, b, c, r, s, list(zip(peds, vals)), challenge))
    assert B.proof_write(proof["raw"], want_cm, proof["pok"]) == cref.proof_write(cref.prove(pk, W, a, b, c, r, s)["raw"], want_cm, want_pok)
    for p in peds:
        c0.pedersen_pk_free(p)
    c0.pk_free(pkh)
    pool.close()
