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
