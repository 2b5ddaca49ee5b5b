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
