"""ctypes loader for oracle/libgroth16_ref.so (the C restatement).

TEST INFRASTRUCTURE ONLY — see oracle/pyref.py.  Arrays are numpy uint64 with the C-ABI layouts
of include/mi355x_groth16.h: Fr/Fp (n,4); G1 affine (n,8); G1 jac (12,); G2 affine (n,16); G2 jac (24,).
"""
from __future__ import annotations
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
WAIT_POLICY = "passive"   # OMP_WAIT_POLICY default for the oracle's OpenMP runtime (None = leave libgomp's own default); see lib()


class PkDesc(C.Structure):
    """Mirror of mi_pk_desc (include/mi355x_groth16.h)."""
    _fields_ = [
        ("log_n", C.c_uint32), ("nb_public", C.c_uint32), ("nb_wires", C.c_uint64),
        ("g1_a", C.c_void_p), ("n_g1_a", C.c_uint64),
        ("g1_b", C.c_void_p), ("n_g1_b", C.c_uint64),
        ("g1_k", C.c_void_p), ("n_g1_k", C.c_uint64),
        ("g1_z", C.c_void_p), ("n_g1_z", C.c_uint64),
        ("g2_b", C.c_void_p), ("n_g2_b", C.c_uint64),
        ("alpha1", C.c_uint64 * 8), ("beta1", C.c_uint64 * 8), ("delta1", C.c_uint64 * 8),
        ("beta2", C.c_uint64 * 16), ("delta2", C.c_uint64 * 16),
        ("infinity_a", C.c_void_p), ("infinity_b", C.c_void_p),
        ("committed_wires", C.c_void_p), ("n_committed", C.c_uint64),
    ]


NATIVE = False        # bench.py's cpu_baseline leg sets this before the first call: build and load the -march=native library ON THIS MACHINE
BUILD_FLAGS = None    # compiler flags of the library that got loaded (reported in bench.py's cpu_baseline.kind)
PORTABLE_FLAGS = "-O3 -march=x86-64-v2 -fopenmp"   # = oracle/Makefile CFLAGS: the library built in the build container travels to another CPU
NATIVE_FLAGS = "-O3 -march=native -fopenmp"


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libgroth16_ref.so")
    if force or not os.path.exists(so):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def build_native():
    """the same source with -march=native, compiled on the machine that will time it (oracle/Makefile `native`); None when no compiler
    or no writable place is to be had -- the portable library then serves"""
    import tempfile
    for d in (_HERE, tempfile.gettempdir()):
        so = os.path.join(d, "libgroth16_ref_native.so")
        try:
            subprocess.check_call(["gcc"] + NATIVE_FLAGS.split() + ["-fPIC", "-Wno-unused-function", "-shared", "-o", so, os.path.join(_HERE, "groth16_ref.c")],
                                  stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300)
            return so
        except Exception:
            continue
    return None


def lib():
    global _LIB
    if _LIB is None:
        # libgomp sizes its spin-waits by the visible cores, not by a container's CPU quota: under a quota the spinning threads
        # burn it and every barrier then costs a scheduler period (seen: 13 s for a 2^12 prove).  Yield instead.
        # bench.py's cpu_baseline leg sets WAIT_POLICY = None: the timed CPU sample keeps libgomp's default (spinning) waits.
        if WAIT_POLICY:
            os.environ.setdefault("OMP_WAIT_POLICY", WAIT_POLICY)
        global BUILD_FLAGS
        so = build_native() if NATIVE else None
        BUILD_FLAGS = NATIVE_FLAGS if so else PORTABLE_FLAGS
        _LIB = C.CDLL(so or build())
        _LIB.ref_proof_write.restype = C.c_size_t
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


def gen_scalars(n, seed, dist):
    out = np.zeros((n, 4), np.uint64)
    lib().ref_gen_scalars(_p(out), C.c_size_t(n), C.c_uint64(seed), C.c_int(dist))
    return out


def gen_g1(n, seed):
    out = np.zeros((n, 8), np.uint64)
    lib().ref_gen_g1(_p(out), C.c_size_t(n), C.c_uint64(seed))
    return out


def gen_g2(n, seed):
    out = np.zeros((n, 16), np.uint64)
    lib().ref_gen_g2(_p(out), C.c_size_t(n), C.c_uint64(seed))
    return out


def field_op(field, op, x, y=None):
    x = u64(x)
    y = x if y is None else u64(y)
    z = np.zeros_like(x)
    rc = lib().ref_field_op(C.c_int(field), C.c_int(op), _p(z), _p(x), _p(y), C.c_size_t(x.shape[0]))
    assert rc == 0
    return z


def g1_add(a, b):
    out = np.zeros_like(a)
    lib().ref_g1_add(_p(out), _p(u64(a)), _p(u64(b)), C.c_size_t(a.shape[0]))
    return out


def g2_add(a, b):
    out = np.zeros_like(a)
    lib().ref_g2_add(_p(out), _p(u64(a)), _p(u64(b)), C.c_size_t(a.shape[0]))
    return out


def g1_on_curve(p):
    return bool(lib().ref_g1_on_curve(_p(u64(p)), C.c_size_t(p.shape[0])))


def g2_on_curve(p):
    return bool(lib().ref_g2_on_curve(_p(u64(p)), C.c_size_t(p.shape[0])))


def ntt(a, log_n, flags):
    a = u64(a).copy()
    rc = lib().ref_ntt(_p(a), C.c_uint32(log_n), C.c_uint32(flags))
    assert rc == 0
    return a


def compute_h(log_n, a, b, c):
    h = np.zeros((1 << log_n, 4), np.uint64)
    rc = lib().ref_compute_h(C.c_uint32(log_n), _p(u64(a)), _p(u64(b)), _p(u64(c)), C.c_size_t(a.shape[0]), _p(h))
    assert rc == 0
    return h


def msm_g1(pts, sc, flags=0, naive=False):
    out = np.zeros(12, np.uint64)
    f = lib().ref_msm_g1_naive if naive else lib().ref_msm_g1
    assert f(_p(u64(pts)), _p(u64(sc)), C.c_size_t(pts.shape[0]), C.c_uint32(flags), _p(out)) == 0
    return out


def msm_g2(pts, sc, flags=0, naive=False):
    out = np.zeros(24, np.uint64)
    f = lib().ref_msm_g2_naive if naive else lib().ref_msm_g2
    assert f(_p(u64(pts)), _p(u64(sc)), C.c_size_t(pts.shape[0]), C.c_uint32(flags), _p(out)) == 0
    return out


def pedersen_msm(bases, values):
    out = np.zeros(8, np.uint64); bases, values = u64(bases), u64(values)
    assert lib().ref_pedersen_msm(_p(bases), _p(values), C.c_size_t(values.shape[0]), _p(out)) == 0
    return out


def pedersen_fold(points, challenge):
    out = np.zeros(8, np.uint64); points = u64(points)
    assert lib().ref_pedersen_fold(_p(points), C.c_size_t(points.shape[0]), _p(u64(challenge)), _p(out)) == 0
    return out


def batch_scalar_mul(base, scalars, g2=False):
    scalars = u64(scalars); out = np.zeros((scalars.shape[0], 16 if g2 else 8), np.uint64)
    f = lib().ref_batch_scalar_mul_g2 if g2 else lib().ref_batch_scalar_mul_g1
    assert f(_p(u64(base)), _p(scalars), C.c_size_t(scalars.shape[0]), _p(out)) == 0
    return out


def g1_sum(parts):
    out = np.zeros(12, np.uint64)
    parts = u64(parts)
    assert lib().ref_g1_sum(_p(parts), C.c_size_t(parts.shape[0]), _p(out)) == 0
    return out


def make_pk_desc(pk: dict):
    """pk: dict of numpy arrays (see tests/helpers.py:synthetic_pk). Returns (PkDesc, keepalive)."""
    d = PkDesc()
    d.log_n, d.nb_public, d.nb_wires = pk["log_n"], pk["nb_public"], pk["nb_wires"]
    keep = []
    for name in ("g1_a", "g1_b", "g1_k", "g1_z", "g2_b"):
        arr = u64(pk[name]); keep.append(arr)
        setattr(d, name, arr.ctypes.data)
        setattr(d, "n_" + name, arr.shape[0])
    for name, k in (("alpha1", 8), ("beta1", 8), ("delta1", 8), ("beta2", 16), ("delta2", 16)):
        setattr(d, name, (C.c_uint64 * k)(*[int(v) for v in u64(pk[name]).reshape(-1)]))
    ia = np.ascontiguousarray(pk["infinity_a"], dtype=np.uint8); ib = np.ascontiguousarray(pk["infinity_b"], dtype=np.uint8)
    keep += [ia, ib]
    d.infinity_a, d.infinity_b = ia.ctypes.data, ib.ctypes.data
    cw = pk.get("committed_wires")
    if cw is not None and len(cw):
        cw = np.ascontiguousarray(cw, dtype=np.uint32); keep.append(cw)
        d.committed_wires, d.n_committed = cw.ctypes.data, cw.shape[0]
    else:
        d.committed_wires, d.n_committed = None, 0
    return d, keep


def prove(pk: dict, W, a, b, c, r, s, want_h=False):
    d, keep = make_pk_desc(pk)
    out = np.zeros(8 + 16 + 8, np.uint64)
    h = np.zeros((1 << pk["log_n"], 4), np.uint64) if want_h else None
    W, a, b, c, r, s = (u64(x) for x in (W, a, b, c, r, s))
    rc = lib().ref_groth16_prove(C.byref(d), _p(W), C.c_size_t(W.shape[0]), _p(a), _p(b), _p(c),
                                 C.c_size_t(a.shape[0]), _p(r), _p(s), _p(out), _p(h))
    assert rc == 0, rc
    proof = {"ar": out[:8].copy(), "bs": out[8:24].copy(), "krs": out[24:].copy(), "raw": out}
    return (proof, h) if want_h else proof


def proof_write(raw, commitments=None, pok=None):
    n = 0 if commitments is None else commitments.shape[0]
    buf = np.zeros(164 + 32 * n, np.uint8)
    ln = lib().ref_proof_write(_p(u64(raw)), _p(commitments), C.c_uint32(n), _p(pok), _p(buf))
    return bytes(buf[:ln])


def g1_compress(p):
    buf = np.zeros(32, np.uint8); lib().ref_g1_compress(_p(u64(p)), _p(buf)); return bytes(buf)


def g2_compress(p):
    buf = np.zeros(64, np.uint8); lib().ref_g2_compress(_p(u64(p)), _p(buf)); return bytes(buf)


def num_threads():
    return int(lib().ref_num_threads())


# ---- conversions between numpy limb arrays (Montgomery) and pyref integers (canonical) ----
def limbs_to_int(row) -> int:
    return sum(int(row[i]) << (64 * i) for i in range(4))


def int_to_limbs(x: int):
    return [(x >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]


def g1_scalar_mul(p, k_canonical_int):
    out = np.zeros(8, np.uint64); k = np.array(int_to_limbs(k_canonical_int), dtype=np.uint64)
    lib().ref_g1_scalar_mul(_p(out), _p(u64(p)), _p(k)); return out


def g2_scalar_mul(p, k_canonical_int):
    out = np.zeros(16, np.uint64); k = np.array(int_to_limbs(k_canonical_int), dtype=np.uint64)
    lib().ref_g2_scalar_mul(_p(out), _p(u64(p)), _p(k)); return out
