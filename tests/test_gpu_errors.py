"""GPU suite: error paths unwind cleanly.  mi_debug_inject_hip_failure makes the n-th checked HIP call of the library fail;
init must then return an error (never crash, never hand out a half-built context) and the next attempt must succeed."""
import ctypes as C
import numpy as np
import pytest
import cref
from helpers import *
from gpu_common import load_binding

pytestmark = pytest.mark.gpu


def test_init_unwinds_on_every_failing_call():
    B = load_binding()
    lib = B.load()
    failures = 0
    for nth in range(1, 200):
        assert lib.mi_debug_inject_hip_failure(nth) == 0
        h = C.c_void_p()
        rc = lib.mi_init(0, C.byref(h))
        lib.mi_debug_inject_hip_failure(0)
        if rc == 0:          # nth is past the last checked call of init: a complete context
            assert h.value
            assert lib.mi_shutdown(h) == 0
            break
        failures += 1
        assert rc in (-2, -3) and not h.value
    assert failures >= 40, failures   # 6 streams + 36 slot events + 6 pinned buffers + 24 context events are all checked
    c = B.Context(0)                  # and the library still works afterwards
    a = cref.gen_scalars(1 << 10, 1, 0)
    assert np.array_equal(c.ntt(a, 10, 0), cref.ntt(a, 10, 0))
    c.close()


def test_prove_reports_injected_failures_and_recovers():
    B = load_binding()
    c = B.Context(0)
    pk = synthetic_pk(10, 1000, 7, 99)
    W = cref.gen_scalars(1000, 1, 1); a = cref.gen_scalars(1020, 2, 1); b = cref.gen_scalars(1020, 3, 0); cc = cref.field_op(0, 2, a, b)
    r, s = cref.gen_scalars(2, 4, 0)
    pkh = c.pk_load(pk)
    want = B.proof_write(c.prove(pkh, W, a, b, cc, r, s)[0]["raw"])
    for nth in (1, 3, 10, 25, 60):
        assert c.lib.mi_debug_inject_hip_failure(nth) == 0
        try:
            with pytest.raises(B.MiError):
                c.prove(pkh, W, a, b, cc, r, s)
        finally:
            c.lib.mi_debug_inject_hip_failure(0)
        c.sync()
        assert B.proof_write(c.prove(pkh, W, a, b, cc, r, s)[0]["raw"]) == want
    c.pk_free(pkh)
    c.close()


def test_pool_rejects_second_waiter_and_unknown_ticket():
    B = load_binding()
    p = B.Prover(0, 2)
    assert p.lib.mi_prover_wait(p.h, C.c_uint64(12345)) != 0
    assert b"unknown ticket" in p.lib.mi_prover_last_error(p.h)
    p.close()


def test_sharded_prove_and_pool_report_injected_failures_and_recover():
    """fault injection through the multi-threaded paths: a group's sharded prove (one host thread per rank) and the prover pool's
    upload stage + workers must turn an injected HIP failure into an error code -- no hang, no crash -- and work again afterwards"""
    B = load_binding()
    pk = synthetic_pk(12, 4000, 7, 199)
    W = cref.gen_scalars(4000, 1, 1); a = cref.gen_scalars(4090, 2, 1); b = cref.gen_scalars(4090, 3, 0); cc = cref.field_op(0, 2, a, b)
    r, s = cref.gen_scalars(2, 4, 0)
    want = cref.proof_write(cref.prove(pk, W, a, b, cc, r, s)["raw"])
    g = B.Group([0, 0])
    try:
        spk = g.pk_load(pk)
        assert B.proof_write(g.prove(spk, W, a, b, cc, r, s, mode=1)[0]["raw"]) == want
        for nth in (1, 2, 5, 9, 17, 33, 70, 120):
            for mode in (0, 1):
                assert g.lib.mi_debug_inject_hip_failure(nth) == 0
                try:
                    failed = False
                    try:
                        g.prove(spk, W, a, b, cc, r, s, mode=mode)
                    except B.MiError:
                        failed = True
                finally:
                    g.lib.mi_debug_inject_hip_failure(0)
                for i in range(g.n_local):
                    g.ctx(i).sync()
                got, _ = g.prove(spk, W, a, b, cc, r, s, mode=mode)      # whatever happened, the next proof is right
                assert B.proof_write(got["raw"]) == want, (nth, mode, failed)
        g.pk_free(spk)
    finally:
        g.close()
    p = B.Prover(0, 2)
    try:
        c0 = p.ctx(0)
        pkh = c0.pk_load(pk)
        for nth in (1, 3, 8, 20, 50):
            assert p.lib.mi_debug_inject_hip_failure(nth) == 0
            tickets = [p.submit(pkh, W, a, b, cc, r, s) for _ in range(3)]
            results = []
            for t in tickets:
                try:
                    results.append(B.proof_write(p.wait(t)[0]["raw"]))
                except B.MiError:
                    results.append(None)
            p.lib.mi_debug_inject_hip_failure(0)
            assert all(x is None or x == want for x in results), nth     # a proof is either refused or right, never wrong
            t = p.submit(pkh, W, a, b, cc, r, s)
            assert B.proof_write(p.wait(t)[0]["raw"]) == want
        c0.pk_free(pkh)
    finally:
        p.close()


def test_compute_h_over_the_ranks_fails_cleanly_when_its_buffers_cannot_be_had():
    """ADVICE r5 (group.hip compute_h_sharded): the cross-rank tables, exchange vectors, h slice and events of computeH over the ranks are
    reserved in the LOCAL phase, before the ranks agree to enter the collective.  On a FRESH group (nothing reserved yet) every early
    MI-checked HIP call of the first sharded prove is made to fail in turn: the call returns an error on every rank -- no kernel or
    transfer ever runs on a null buffer (that would be a GPU memory fault, not an error code) -- the group stays usable unless a transfer
    itself failed, and the next proof on a usable group is the oracle's."""
    B = load_binding()
    pk = synthetic_pk(12, 4000, 7, 299)
    W = cref.gen_scalars(4000, 1, 1); a = cref.gen_scalars(4090, 2, 1); b = cref.gen_scalars(4090, 3, 0); cc = cref.field_op(0, 2, a, b)
    r, s = cref.gen_scalars(2, 4, 0)
    want = cref.proof_write(cref.prove(pk, W, a, b, cc, r, s)["raw"])
    outcomes = {"failed_usable": 0, "failed_broken": 0, "beyond": 0}
    for nth in list(range(1, 40, 2)) + [48, 64, 90]:
        g = B.Group([0, 0])
        try:
            g.set_sharded_compute_h(True)
            spk = g.pk_load(pk)
            assert g.lib.mi_debug_inject_hip_failure(nth) == 0
            failed = False
            try:
                got, _ = g.prove(spk, W, a, b, None, r, s, mode=1)
            except B.MiError:
                failed = True
            finally:
                g.lib.mi_debug_inject_hip_failure(0)
            for i in range(g.n_local):
                g.ctx(i).sync()
            if not failed:
                assert B.proof_write(got["raw"]) == want, nth
                outcomes["beyond"] += 1
                continue
            try:
                got, _ = g.prove(spk, W, a, b, None, r, s, mode=1)
                assert B.proof_write(got["raw"]) == want, nth
                outcomes["failed_usable"] += 1
            except B.MiError as e:     # a failure INSIDE an exchange breaks the group by contract: every later call is refused with that message
                assert "destroy the group" in str(e) or "failed" in str(e), str(e)
                outcomes["failed_broken"] += 1
        finally:
            g.close()
    assert outcomes["failed_usable"] >= 5, outcomes   # the local phase (reserves included) is where most early calls are
