"""ctypes binding of libmi355x_groth16.so (the C-ABI of include/mi355x_groth16.h).

This is plumbing for tests/ and bench.py: it mirrors the C entry points one to one and adds no
compute.  There is NO fallback: if the HIP library is missing or no gfx950 device is present,
load()/Context() raise.  numpy layouts: Fr/Fp (n,4) uint64; G1 affine (n,8); G1 jac (12,);
G2 affine (n,16); G2 jac (24,), all Montgomery little-endian limbs.
"""
from __future__ import annotations
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MI355X_GROTH16_LIB", os.path.join(_HERE, "libmi355x_groth16.so"))   # the override is for A/B runs of two builds
_LIB = None

EXPORTS = [
    "mi_init", "mi_init_prio", "mi_shutdown", "mi_last_error", "mi_set_stream", "mi_pk_load", "mi_pk_load_dev", "mi_pk_free",
    "mi_ntt", "mi_ntt_dev", "mi_compute_h", "mi_compute_h_dev", "mi_msm_g1", "mi_msm_g1_dev", "mi_msm_g2",
    "mi_msm_g2_dev", "mi_groth16_prove", "mi_groth16_prove_dev", "mi_get_stats", "mi_g1_compress",
    "mi_g2_compress", "mi_proof_write", "mi_g1_sum", "mi_g2_sum", "mi_gen_scalars_dev", "mi_gen_g1_dev",
    "mi_gen_g2_dev", "mi_field_op_dev", "mi_g1_add_dev", "mi_g2_add_dev", "mi_bench_modmul_dev", "mi_bench_valu_dev", "mi_bench_gather_dev",
    "mi_dev_alloc", "mi_dev_free", "mi_dev_upload", "mi_dev_download", "mi_dev_sync",
    "mi_msm_precompute_g1_dev", "mi_msm_precompute_g2_dev", "mi_msm_g1_fixed_dev", "mi_msm_g2_fixed_dev", "mi_msm_table_to_rprime_g1_dev", "mi_msm_table_to_rprime_g2_dev", "mi_pk_table_plan",
    "mi_batch_scalar_mul_g1", "mi_batch_scalar_mul_g1_dev", "mi_batch_scalar_mul_g2", "mi_batch_scalar_mul_g2_dev",
    "mi_pedersen_pk_load", "mi_pedersen_pk_free", "mi_pedersen_commit", "mi_pedersen_prove_knowledge", "mi_pedersen_fold",
    "mi_debug_set_prove_fixed_base", "mi_debug_set_prove_schedule", "mi_debug_set_msm_batch_affine", "mi_debug_set_msm_group_bits", "mi_debug_inject_hip_failure", "mi_debug_set_ntt_plan", "mi_debug_set_ntt_threads", "mi_debug_set_ntt_wave_stages",
    "mi_debug_set_msm_plan", "mi_debug_set_msm_chunk", "mi_debug_set_msm_one_pass_sort", "mi_debug_set_msm_bound_levels", "mi_debug_set_msm_limb29", "mi_debug_set_msm_l1_waves", "mi_debug_set_msm_precompute_batched", "mi_debug_set_ntt_fuse_pair", "mi_debug_set_knob", "mi_debug_get_counter", "mi_debug_set_stream_plan", "mi_debug_set_trace_ranges", "mi_set_trace_ranges",
    "mi_prover_create", "mi_prover_destroy", "mi_prover_in_flight", "mi_prover_ctx", "mi_prover_last_error",
    "mi_prover_submit", "mi_prover_submit_dev", "mi_prover_wait", "mi_prover_commit", "mi_prover_submit_bsb22", "mi_prover_trim", "mi_ctx_trim",
    "mi_group_create", "mi_group_unique_id", "mi_group_create_rank", "mi_group_create_rank_ex", "mi_group_rank", "mi_group_destroy", "mi_group_world", "mi_group_local", "mi_group_ctx",
    "mi_group_last_error", "mi_group_transport", "mi_group_comm_ranks", "mi_group_device_pci", "mi_group_set_lead_share", "mi_group_wire_range", "mi_group_set_sharded_compute_h", "mi_compute_h_sharded_dev", "mi_groth16_prove_sharded_slices_dev", "mi_group_exchange_selftest", "mi_pk_load_sharded", "mi_pk_sharded_free",
    "mi_groth16_prove_sharded", "mi_groth16_prove_sharded_dev", "mi_pk_load_sharded_dev", "mi_msm_g1_sharded", "mi_msm_g2_sharded",
    "mi_msm_g1_sharded_dev", "mi_msm_g2_sharded_dev",
    "mi_pk_raw_inspect", "mi_pk_load_raw",
    "mi_whir_proof_decode", "mi_whir_proof_free", "mi_whir_proof_elements", "mi_whir_proof_statement_values", "mi_whir_element_shape", "mi_whir_parse_paths",
    "mi_whir_reverse", "mi_whir_prefix_decode_path", "mi_whir_limbs_to_fr", "mi_whir_interner_decode", "mi_whir_matrix_cells", "mi_whir_config_parse", "mi_whir_config_free",
]


class PkDesc(C.Structure):
    _fields_ = [
        ("log_n", C.c_uint32), ("nb_public", C.c_uint32), ("nb_wires", C.c_uint64),
        ("g1_a", C.c_void_p), ("n_g1_a", C.c_uint64), ("g1_b", C.c_void_p), ("n_g1_b", C.c_uint64),
        ("g1_k", C.c_void_p), ("n_g1_k", C.c_uint64), ("g1_z", C.c_void_p), ("n_g1_z", C.c_uint64),
        ("g2_b", C.c_void_p), ("n_g2_b", C.c_uint64),
        ("alpha1", C.c_uint64 * 8), ("beta1", C.c_uint64 * 8), ("delta1", C.c_uint64 * 8),
        ("beta2", C.c_uint64 * 16), ("delta2", C.c_uint64 * 16),
        ("infinity_a", C.c_void_p), ("infinity_b", C.c_void_p),
        ("committed_wires", C.c_void_p), ("n_committed", C.c_uint64),
    ]


class MemLedger(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("key_bases", "key_tables", "key_indices", "ctx_ntt_tables", "ctx_ntt_vectors", "ctx_msm", "ctx_other")]


class PkRawInfo(C.Structure):
    _fields_ = [("log_n", C.c_uint32), ("n_commitment_keys", C.c_uint32)] + [(n, C.c_uint64) for n in (
        "nb_wires", "n_g1_a", "n_g1_b", "n_g1_z", "n_g1_k", "n_g2_b", "off_alpha1", "off_g1_a", "off_g1_b", "off_g1_z", "off_g1_k",
        "off_beta2", "off_g2_b", "off_infinity_a", "off_infinity_b")] + [
        ("n_basis", C.c_uint64 * 16), ("off_basis", C.c_uint64 * 16), ("off_basis_exp_sigma", C.c_uint64 * 16)]


def pk_raw_inspect(blob: bytes):
    """section offsets / counts of a gnark ProvingKey.WriteRawTo stream (host only); raises MiError if the layout does not parse"""
    info = PkRawInfo(); buf = (C.c_uint8 * len(blob)).from_buffer_copy(blob)
    if load().mi_pk_raw_inspect(buf, C.c_size_t(len(blob)), C.byref(info)) != 0:
        raise MiError("mi_pk_raw_inspect: not a gnark v0.11.0 ProvingKey.WriteRawTo stream (as recalled)")
    return info


class Bsb22Input(C.Structure):
    _fields_ = [("key", C.c_void_p), ("values", C.c_void_p), ("n", C.c_size_t)]


class Stats(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("h2d_ms", "compute_h_ms", "filter_ms", "msm_a_ms", "msm_b1_ms", "msm_k_ms",
                                         "msm_z_ms", "msm_b2_ms", "assemble_ms", "total_ms")] + [
        ("g1_accum_kernel_ms", C.c_float), ("g1_accum_pairs", C.c_uint64), ("g1_accum_launches", C.c_uint32),
        ("ntt_kernel_ms", C.c_float), ("ntt_elems", C.c_uint64), ("ntt_launches", C.c_uint32), ("g1_accum_entries", C.c_uint64), ("g1_level1_additions", C.c_uint64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class MiError(RuntimeError):
    pass


def load():
    """dlopen the HIP library; raises if it has not been built (no CPU fallback exists)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise MiError(f"{LIB_PATH} missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        _LIB = C.CDLL(LIB_PATH)
        _LIB.mi_last_error.restype = C.c_char_p
        _LIB.mi_proof_write.restype = C.c_size_t
        _LIB.mi_prover_last_error.restype = C.c_char_p
        _LIB.mi_prover_ctx.restype = C.c_void_p
        _LIB.mi_prover_in_flight.restype = C.c_uint32
        _LIB.mi_group_ctx.restype = C.c_void_p
        _LIB.mi_group_last_error.restype = C.c_char_p
    return _LIB


DIST_UNIFORM, DIST_WHIR = 0, 1


def dist_mix(bit_pm, byte_pm, u64_pm=0):
    """MI_DIST_MIX(bit, byte, u64) of include/mi355x_groth16_debug.h: per-mille shares of {0,1} / bytes / 64-bit scalars, the rest uniform Fr"""
    assert 0 <= bit_pm and 0 <= byte_pm and 0 <= u64_pm and bit_pm + byte_pm + u64_pm <= 1000
    return 0x40000000 | (bit_pm << 20) | (byte_pm << 10) | u64_pm


def _p(a):
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data_as(C.c_void_p)
    return C.c_void_p(int(a))  # raw device pointer


def _u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


class DevArray:
    """A hipMalloc'ed buffer owned through the C-ABI (mi_dev_alloc)."""

    def __init__(self, ctx, nbytes):
        self.ctx, self.nbytes = ctx, int(nbytes)
        ptr = C.c_void_p()
        ctx._ck(ctx.lib.mi_dev_alloc(ctx.h, C.c_size_t(self.nbytes), C.byref(ptr)))
        self.ptr = ptr.value

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        self.ctx._ck(self.ctx.lib.mi_dev_upload(self.ctx.h, C.c_void_p(self.ptr), _p(arr), C.c_size_t(arr.nbytes)))
        return self

    def download(self, shape, dtype=np.uint64):
        out = np.zeros(shape, dtype)
        assert out.nbytes <= self.nbytes
        self.ctx._ck(self.ctx.lib.mi_dev_download(self.ctx.h, _p(out), C.c_void_p(self.ptr), C.c_size_t(out.nbytes)))
        return out

    def free(self):
        if self.ptr:
            self.ctx.lib.mi_dev_free(self.ctx.h, C.c_void_p(self.ptr))
            self.ptr = None


class Context:
    def __init__(self, device_id=0, _borrowed=None):
        self.lib = load()
        self.owned = _borrowed is None
        if _borrowed is not None:      # a context owned by a Prover pool (mi_prover_ctx)
            self.h = C.c_void_p(_borrowed)
            return
        h = C.c_void_p()
        rc = self.lib.mi_init(C.c_int(device_id), C.byref(h))
        if rc != 0:
            raise MiError(f"mi_init failed: {rc} (no gfx950 device? the product has no CPU path)")
        self.h = h

    def _ck(self, rc):
        if rc != 0:
            raise MiError(f"rc={rc}: {self.lib.mi_last_error(self.h).decode()}")

    def close(self):
        if self.h and self.owned:
            self.lib.mi_shutdown(self.h)
        self.h = None

    def set_stream(self, stream_ptr):
        self._ck(self.lib.mi_set_stream(self.h, C.c_void_p(int(stream_ptr))))

    def sync(self):
        self._ck(self.lib.mi_dev_sync(self.h))

    def set_knob(self, name, value):
        """mi_debug_set_knob: a named measurement / test knob of this context (the header lists them)"""
        self._ck(self.lib.mi_debug_set_knob(self.h, C.c_char_p(name.encode()), C.c_int64(int(value))))

    def counter(self, name):
        """mi_debug_get_counter: how often an optional path ran on this context"""
        v = C.c_uint64()
        self._ck(self.lib.mi_debug_get_counter(self.h, C.c_char_p(name.encode()), C.byref(v)))
        return int(v.value)

    def alloc(self, nbytes):
        return DevArray(self, nbytes)

    def to_dev(self, arr):
        arr = np.ascontiguousarray(arr)
        return DevArray(self, max(arr.nbytes, 32)).upload(arr)

    # ---- utilities
    def gen_scalars(self, n, seed, dist):
        d = self.alloc(32 * n); self._ck(self.lib.mi_gen_scalars_dev(self.h, _p(d.ptr), C.c_size_t(n), C.c_uint64(seed), C.c_int(dist))); return d

    def gen_g1(self, n, seed):
        d = self.alloc(64 * n); self._ck(self.lib.mi_gen_g1_dev(self.h, _p(d.ptr), C.c_size_t(n), C.c_uint64(seed))); return d

    def gen_g2(self, n, seed):
        d = self.alloc(128 * n); self._ck(self.lib.mi_gen_g2_dev(self.h, _p(d.ptr), C.c_size_t(n), C.c_uint64(seed))); return d

    def field_op(self, field, op, x, y=None):
        x = _u64(x); n = x.shape[0]
        dx = self.to_dev(x); dy = self.to_dev(_u64(y)) if y is not None else None; dz = self.alloc(32 * n)
        self._ck(self.lib.mi_field_op_dev(self.h, field, op, _p(dz.ptr), _p(dx.ptr), _p(dy.ptr if dy else None), C.c_size_t(n)))
        out = dz.download((n, 4))
        for b in (dx, dy, dz):
            if b: b.free()
        return out

    def field_op_dev(self, field, op, z_ptr, x_ptr, y_ptr, n):
        self._ck(self.lib.mi_field_op_dev(self.h, C.c_int(field), C.c_int(op), _p(z_ptr), _p(x_ptr), _p(y_ptr), C.c_size_t(n)))

    def ec_add(self, a, b, g2=False):
        a, b = _u64(a), _u64(b); n = a.shape[0]
        da, db = self.to_dev(a), self.to_dev(b); do = self.alloc(a.nbytes)
        f = self.lib.mi_g2_add_dev if g2 else self.lib.mi_g1_add_dev
        self._ck(f(self.h, _p(do.ptr), _p(da.ptr), _p(db.ptr), C.c_size_t(n)))
        out = do.download(a.shape)
        for x in (da, db, do): x.free()
        return out

    def bench_modmul(self, field, n_threads, iters):
        s = self.alloc(1024 * 32); ms = C.c_float()
        self._ck(self.lib.mi_bench_modmul_dev(self.h, field, C.c_size_t(n_threads), C.c_uint32(iters), _p(s.ptr), C.byref(ms)))
        s.free(); return ms.value

    def bench_gather(self, table_ptr, n_entries, n_threads, iters):
        s = self.alloc(1024); ms = C.c_float()
        self._ck(self.lib.mi_bench_gather_dev(self.h, _p(table_ptr), C.c_size_t(n_entries), C.c_size_t(n_threads), C.c_uint32(iters), _p(s.ptr), C.byref(ms)))
        s.free(); return ms.value

    def bench_valu(self, kind, n_threads, iters):
        s = self.alloc(1024); ms = C.c_float()
        self._ck(self.lib.mi_bench_valu_dev(self.h, kind, C.c_size_t(n_threads), C.c_uint32(iters), _p(s.ptr), C.byref(ms)))
        s.free(); return ms.value

    # ---- NTT / computeH
    def ntt(self, a, log_n, flags):
        a = _u64(a).copy(); self._ck(self.lib.mi_ntt(self.h, _p(a), C.c_uint32(log_n), C.c_uint32(flags))); return a

    def ntt_dev(self, dptr, log_n, flags):
        self._ck(self.lib.mi_ntt_dev(self.h, _p(dptr), C.c_uint32(log_n), C.c_uint32(flags)))

    def compute_h(self, log_n, a, b, c):
        h = np.zeros((1 << log_n, 4), np.uint64)
        a, b, c = _u64(a), _u64(b), (None if c is None else _u64(c))
        self._ck(self.lib.mi_compute_h(self.h, C.c_uint32(log_n), _p(a), _p(b), _p(c), C.c_size_t(a.shape[0]), _p(h)))
        return h

    def compute_h_dev(self, log_n, a, b, c, n_constraints, h):
        self._ck(self.lib.mi_compute_h_dev(self.h, C.c_uint32(log_n), _p(a), _p(b), _p(c), C.c_size_t(n_constraints), _p(h)))

    # ---- MSM
    def msm_g1(self, pts, sc, flags=0):
        out = np.zeros(12, np.uint64); pts, sc = _u64(pts), _u64(sc)
        self._ck(self.lib.mi_msm_g1(self.h, _p(pts), _p(sc), C.c_size_t(pts.shape[0]), C.c_uint32(flags), _p(out))); return out

    def msm_g2(self, pts, sc, flags=0):
        out = np.zeros(24, np.uint64); pts, sc = _u64(pts), _u64(sc)
        self._ck(self.lib.mi_msm_g2(self.h, _p(pts), _p(sc), C.c_size_t(pts.shape[0]), C.c_uint32(flags), _p(out))); return out

    def msm_g1_dev(self, pts_ptr, sc_ptr, n, flags=0):
        out = np.zeros(12, np.uint64)
        self._ck(self.lib.mi_msm_g1_dev(self.h, _p(pts_ptr), _p(sc_ptr), C.c_size_t(n), C.c_uint32(flags), _p(out))); return out

    def msm_g2_dev(self, pts_ptr, sc_ptr, n, flags=0):
        out = np.zeros(24, np.uint64)
        self._ck(self.lib.mi_msm_g2_dev(self.h, _p(pts_ptr), _p(sc_ptr), C.c_size_t(n), C.c_uint32(flags), _p(out))); return out

    # ---- fixed-base MSM (window copies of static bases)
    def msm_precompute(self, base_ptr, n, c, g2=False):
        nwin = (256 + c - 1) // c
        pre = self.alloc((128 if g2 else 64) * n * nwin)
        f = self.lib.mi_msm_precompute_g2_dev if g2 else self.lib.mi_msm_precompute_g1_dev
        self._ck(f(self.h, _p(base_ptr), C.c_size_t(n), C.c_uint32(c), _p(pre.ptr)))
        return pre

    def msm_table_to_rprime(self, pre_ptr, n_points, g2=False):
        f = self.lib.mi_msm_table_to_rprime_g2_dev if g2 else self.lib.mi_msm_table_to_rprime_g1_dev
        self._ck(f(self.h, _p(pre_ptr), C.c_size_t(n_points)))

    def pk_table_plan(self, pkh):
        c = (C.c_uint32 * 3)()
        self._ck(self.lib.mi_pk_table_plan(pkh, c))
        return tuple(int(x) for x in c)

    def msm_fixed_dev(self, pre_ptr, sc_ptr, n, c, flags=0, g2=False):
        out = np.zeros(24 if g2 else 12, np.uint64)
        f = self.lib.mi_msm_g2_fixed_dev if g2 else self.lib.mi_msm_g1_fixed_dev
        self._ck(f(self.h, _p(pre_ptr), _p(sc_ptr), C.c_size_t(n), C.c_uint32(c), C.c_uint32(flags), _p(out)))
        return out

    # ---- proving key + prove
    def pk_load(self, pk: dict, device_points=False):
        """pk arrays: numpy (host) or raw device pointers with explicit counts when device_points."""
        d = PkDesc(); keep = []
        d.log_n, d.nb_public, d.nb_wires = pk["log_n"], pk["nb_public"], pk["nb_wires"]
        for name in ("g1_a", "g1_b", "g1_k", "g1_z", "g2_b"):
            if device_points:
                ptr, cnt = pk[name]
                setattr(d, name, int(ptr)); setattr(d, "n_" + name, int(cnt))
            else:
                arr = _u64(pk[name]); keep.append(arr)
                setattr(d, name, arr.ctypes.data); setattr(d, "n_" + name, arr.shape[0])
        for name, k in (("alpha1", 8), ("beta1", 8), ("delta1", 8), ("beta2", 16), ("delta2", 16)):
            setattr(d, name, (C.c_uint64 * k)(*[int(v) for v in _u64(pk[name]).reshape(-1)]))
        ia = np.ascontiguousarray(pk["infinity_a"], dtype=np.uint8); ib = np.ascontiguousarray(pk["infinity_b"], dtype=np.uint8)
        keep += [ia, ib]
        d.infinity_a, d.infinity_b = ia.ctypes.data, ib.ctypes.data
        cw = pk.get("committed_wires")
        if cw is not None and len(cw):
            cw = np.ascontiguousarray(cw, dtype=np.uint32); keep.append(cw)
            d.committed_wires, d.n_committed = cw.ctypes.data, cw.shape[0]
        h = C.c_void_p()
        f = self.lib.mi_pk_load_dev if device_points else self.lib.mi_pk_load
        self._ck(f(self.h, C.byref(d), C.byref(h)))
        return h

    def pk_load_raw(self, blob: bytes, nb_public, committed_wires=None):
        """gnark ProvingKey.WriteRawTo stream -> (device-resident key, [Pedersen key handles])"""
        buf = (C.c_uint8 * len(blob)).from_buffer_copy(blob); h = C.c_void_p()
        ped = (C.c_void_p * 16)(); nped = C.c_uint32()
        cw = None if committed_wires is None or not len(committed_wires) else np.ascontiguousarray(committed_wires, dtype=np.uint32)
        self._ck(self.lib.mi_pk_load_raw(self.h, buf, C.c_size_t(len(blob)), C.c_uint32(nb_public), _p(cw), C.c_size_t(0 if cw is None else cw.shape[0]),
                                         C.byref(h), ped, C.byref(nped)))
        return h, [C.c_void_p(ped[i]) for i in range(nped.value)]

    def pk_free(self, pkh):
        self._ck(self.lib.mi_pk_free(self.h, pkh))

    def prove(self, pkh, W, a, b, c, r, s, device=False, n_wires=None, n_constraints=None):
        out = np.zeros(32, np.uint64); st = Stats()
        r, s = _u64(r), _u64(s)
        if device:
            self._ck(self.lib.mi_groth16_prove_dev(self.h, pkh, _p(W), C.c_size_t(n_wires), _p(a), _p(b), _p(c),
                                                   C.c_size_t(n_constraints), _p(r), _p(s), _p(out), C.byref(st)))
        else:
            W, a, b, c = _u64(W), _u64(a), _u64(b), (None if c is None else _u64(c))   # c = None: formed on the device as a o b
            self._ck(self.lib.mi_groth16_prove(self.h, pkh, _p(W), C.c_size_t(W.shape[0]), _p(a), _p(b), _p(c),
                                               C.c_size_t(a.shape[0]), _p(r), _p(s), _p(out), C.byref(st)))
        return {"ar": out[:8].copy(), "bs": out[8:24].copy(), "krs": out[24:].copy(), "raw": out}, st.as_dict()

    # ---- fixed-base batch scalar multiplication
    def batch_scalar_mul(self, base, scalars, g2=False):
        base, scalars = _u64(base), _u64(scalars); out = np.zeros((scalars.shape[0], 16 if g2 else 8), np.uint64)
        f = self.lib.mi_batch_scalar_mul_g2 if g2 else self.lib.mi_batch_scalar_mul_g1
        self._ck(f(self.h, _p(base), _p(scalars), C.c_size_t(scalars.shape[0]), _p(out)))
        return out

    def batch_scalar_mul_dev(self, base, scalars_ptr, n, out_ptr, g2=False):
        f = self.lib.mi_batch_scalar_mul_g2_dev if g2 else self.lib.mi_batch_scalar_mul_g1_dev
        self._ck(f(self.h, _p(_u64(base)), _p(scalars_ptr), C.c_size_t(n), _p(out_ptr)))

    # ---- BSB22 Pedersen key
    def pedersen_pk_load(self, basis, basis_exp_sigma):
        basis, bes = _u64(basis), _u64(basis_exp_sigma); h = C.c_void_p()
        self._ck(self.lib.mi_pedersen_pk_load(self.h, _p(basis), _p(bes), C.c_size_t(basis.shape[0]), C.byref(h)))
        return h

    def pedersen_pk_free(self, pk):
        self._ck(self.lib.mi_pedersen_pk_free(self.h, pk))

    def pedersen_commit(self, pk, values, knowledge=False):
        values = _u64(values); out = np.zeros(8, np.uint64)
        f = self.lib.mi_pedersen_prove_knowledge if knowledge else self.lib.mi_pedersen_commit
        self._ck(f(self.h, pk, _p(values), C.c_size_t(values.shape[0]), _p(out)))
        return out

    def trim(self):
        """mi_ctx_trim: every grow-only workspace of this (idle) context goes back to the device"""
        self._ck(self.lib.mi_ctx_trim(self.h))

    def stats(self):
        st = Stats(); self._ck(self.lib.mi_get_stats(self.h, C.byref(st))); return st.as_dict()

    def mem_ledger(self, pkh=None):
        """device memory held by the key `pkh` and by this context, in GB (mi_get_mem_ledger)"""
        m = MemLedger(); self._ck(self.lib.mi_get_mem_ledger(self.h, pkh, C.byref(m)))
        return {n: getattr(m, n) / 1e9 for n, _ in MemLedger._fields_}


# ---- host-only helpers (no ctx)
class Prover:
    """Several proofs in flight on one device (mi_prover_*, include/mi355x_groth16.h).  Mirrors a prover service that
    calls groth16.Prove (reference mt.go:496) from many goroutines: submit() returns a ticket, wait() the proof."""

    def __init__(self, device_id=0, in_flight=2):
        self.lib = load()
        h = C.c_void_p()
        rc = self.lib.mi_prover_create(C.c_int(device_id), C.c_uint32(in_flight), C.byref(h))
        if rc != 0:
            raise MiError(f"mi_prover_create failed: {rc} (no gfx950 device? the product has no CPU path)")
        self.h = h
        self.in_flight = int(self.lib.mi_prover_in_flight(h))
        self._pending = {}

    def ctx(self, i=0) -> "Context":
        """context i of the pool, for mi_pk_load / device buffers (never prove on it directly)"""
        p = self.lib.mi_prover_ctx(self.h, C.c_uint32(i))
        if not p:
            raise MiError("mi_prover_ctx: index out of range")
        return Context(_borrowed=p)

    def set_knob(self, name, value):
        """the knob on every proving context of the pool (call while the pool is idle)"""
        for i in range(self.in_flight):
            self.ctx(i).set_knob(name, value)

    def submit(self, pkh, W, a, b, c, r, s, device=False, n_wires=None, n_constraints=None) -> int:
        out = np.zeros(32, np.uint64); st = Stats(); t = C.c_uint64()
        r, s = _u64(r), _u64(s)
        if device:
            keep = ()
            rc = self.lib.mi_prover_submit_dev(self.h, pkh, _p(W), C.c_size_t(n_wires), _p(a), _p(b), _p(c), C.c_size_t(n_constraints),
                                               _p(r), _p(s), _p(out), C.byref(st), C.byref(t))
        else:
            W, a, b, c = _u64(W), _u64(a), _u64(b), (None if c is None else _u64(c))   # c = None: formed on the device as a o b
            keep = (W, a, b, c)
            rc = self.lib.mi_prover_submit(self.h, pkh, _p(W), C.c_size_t(W.shape[0]), _p(a), _p(b), _p(c), C.c_size_t(a.shape[0]),
                                           _p(r), _p(s), _p(out), C.byref(st), C.byref(t))
        if rc != 0:
            raise MiError(f"mi_prover_submit: rc={rc}")
        self._pending[t.value] = (out, st, keep)   # the library writes into these until wait() returns
        return t.value

    def wait(self, ticket):
        out, st, keep = self._pending.pop(ticket)
        rc = self.lib.mi_prover_wait(self.h, C.c_uint64(ticket))
        if rc != 0:
            raise MiError(f"rc={rc}: {self.lib.mi_prover_last_error(self.h).decode()}")
        res = {"ar": out[:8].copy(), "bs": out[8:24].copy(), "krs": out[24:].copy(), "raw": out}
        if isinstance(keep, dict) and "pok" in keep:
            res["pok"] = keep["pok"]
        return res, st.as_dict()

    # ---- BSB22 through the pool (mi_prover_commit / mi_prover_submit_bsb22)
    def commit(self, ped_pk, values):
        """pedersen Commit inside the solve: synchronous, thread-safe"""
        values = _u64(values); out = np.zeros(8, np.uint64)
        rc = self.lib.mi_prover_commit(self.h, ped_pk, _p(values), C.c_size_t(values.shape[0]), _p(out))
        if rc != 0:
            raise MiError(f"mi_prover_commit rc={rc}: {self.lib.mi_prover_last_error(self.h).decode()}")
        return out

    def submit_bsb22(self, pkh, W, a, b, c, r, s, commitments, challenge) -> int:
        """host inputs; commitments = [(pedersen key handle, private committed values)]; wait() returns the proof with res['pok']"""
        out = np.zeros(32, np.uint64); st = Stats(); t = C.c_uint64(); pok = np.zeros(8, np.uint64)
        r, s, challenge = _u64(r), _u64(s), _u64(challenge)
        W, a, b, c = _u64(W), _u64(a), _u64(b), (None if c is None else _u64(c))
        vals = [_u64(v) for _, v in commitments]
        arr = (Bsb22Input * len(commitments))()
        for i, ((key, _), v) in enumerate(zip(commitments, vals)):
            arr[i].key = key if isinstance(key, int) else key.value; arr[i].values = v.ctypes.data; arr[i].n = v.shape[0]
        rc = self.lib.mi_prover_submit_bsb22(self.h, pkh, _p(W), C.c_size_t(W.shape[0]), _p(a), _p(b), _p(c), C.c_size_t(a.shape[0]),
                                             _p(r), _p(s), arr, C.c_uint32(len(commitments)), _p(challenge), _p(out), _p(pok), C.byref(st), C.byref(t))
        if rc != 0:
            raise MiError(f"mi_prover_submit_bsb22: rc={rc}")
        self._pending[t.value] = (out, st, {"keep": (W, a, b, c, vals, arr, challenge), "pok": pok})
        return t.value

    def trim(self):
        rc = self.lib.mi_prover_trim(self.h)
        if rc != 0:
            raise MiError(f"mi_prover_trim rc={rc}: {self.lib.mi_prover_last_error(self.h).decode()}")

    def close(self):
        if self.h:
            self.lib.mi_prover_destroy(self.h)
            self.h = None


def shard_range(total, world, rank):
    """slice [lo, hi) of rank `rank`: the same cut as the library's (csrc/group.hip range_of)"""
    return total * rank // world, total * (rank + 1) // world


def _fill_pk_desc(pk: dict):
    """whole-key descriptor over HOST arrays (mi_pk_load / mi_pk_load_sharded)"""
    d = PkDesc(); keep = []
    d.log_n, d.nb_public, d.nb_wires = pk["log_n"], pk["nb_public"], pk["nb_wires"]
    for name in ("g1_a", "g1_b", "g1_k", "g1_z", "g2_b"):
        arr = _u64(pk[name]); keep.append(arr)
        setattr(d, name, arr.ctypes.data); setattr(d, "n_" + name, arr.shape[0])
    for name, k in (("alpha1", 8), ("beta1", 8), ("delta1", 8), ("beta2", 16), ("delta2", 16)):
        setattr(d, name, (C.c_uint64 * k)(*[int(v) for v in _u64(pk[name]).reshape(-1)]))
    ia = np.ascontiguousarray(pk["infinity_a"], dtype=np.uint8); ib = np.ascontiguousarray(pk["infinity_b"], dtype=np.uint8)
    keep += [ia, ib]
    d.infinity_a, d.infinity_b = ia.ctypes.data, ib.ctypes.data
    cw = pk.get("committed_wires")
    if cw is not None and len(cw):
        cw = np.ascontiguousarray(cw, dtype=np.uint32); keep.append(cw)
        d.committed_wires, d.n_committed = cw.ctypes.data, cw.shape[0]
    return d, keep


class Group:
    """Several GPUs behind one handle (mi_group_*, include/mi355x_groth16.h): the point-sharded prove and MSM of SURVEY 8e.
    Group([0, 1, ...]) = all ranks in this process (what a Go caller uses); Group.rank(dev, rank, world, id) = one rank per
    process (torch.distributed.run): the 128-byte id comes from Group.unique_id() on rank 0 through the caller's own channel."""

    def __init__(self, dev_ids=None, _h=None):
        self.lib = load()
        if _h is not None:
            self.h = _h
        else:
            arr = (C.c_int * len(dev_ids))(*dev_ids); h = C.c_void_p()
            rc = self.lib.mi_group_create(arr, C.c_int(len(dev_ids)), C.byref(h))
            if rc != 0:
                raise MiError(f"mi_group_create failed: {rc} (no gfx950 device? the product has no CPU path)")
            self.h = h
        self.world = int(self.lib.mi_group_world(self.h)); self.n_local = int(self.lib.mi_group_local(self.h))

    @staticmethod
    def unique_id() -> bytes:
        buf = (C.c_uint8 * 128)()
        if load().mi_group_unique_id(buf) != 0:
            raise MiError("mi_group_unique_id failed")
        return bytes(buf)

    @classmethod
    def rank(cls, device_id, rank, world, uid: bytes, transport=1):
        """transport 1 = RCCL (uid from unique_id()), 3 = host-staged through shared memory (uid = any 128 bytes the ranks share)"""
        h = C.c_void_p(); buf = (C.c_uint8 * 128)(*uid)
        rc = load().mi_group_create_rank_ex(C.c_int(device_id), C.c_int(rank), C.c_int(world), buf, C.c_int(transport), C.byref(h))
        if rc != 0:
            raise MiError(f"mi_group_create_rank_ex failed: {rc}")
        return cls(_h=h)

    def _ck(self, rc):
        if rc != 0:
            raise MiError(f"rc={rc}: {self.lib.mi_group_last_error(self.h).decode()}")

    def ctx(self, i=0) -> "Context":
        p = self.lib.mi_group_ctx(self.h, C.c_int(i))
        if not p:
            raise MiError("mi_group_ctx: index out of range")
        return Context(_borrowed=p)

    def transport(self):
        return {1: "rccl", 2: "peer-copy", 3: "host-staged"}[int(self.lib.mi_group_transport(self.h))]

    def rank_index(self):
        return int(self.lib.mi_group_rank(self.h))

    def comm_ranks(self):
        """ranks the RCCL communicator itself reports (ncclCommCount); 0 = the transport is not RCCL"""
        n = int(self.lib.mi_group_comm_ranks(self.h))
        if n < 0:
            raise MiError(f"mi_group_comm_ranks: {n}")
        return n

    def device_pci(self, local_rank=0):
        buf = C.create_string_buffer(32)
        self._ck(self.lib.mi_group_device_pci(self.h, C.c_int(local_rank), buf))
        return buf.value.decode()

    def set_lead_share(self, permille=0xFFFFFFFF):
        """rank 0's share of the wires, permille of an even share (0xFFFFFFFF = automatic); before pk_load*, the same on every rank"""
        self._ck(self.lib.mi_group_set_lead_share(self.h, C.c_uint32(permille)))

    def wire_range(self, nb_wires, rank):
        """wires [lo, hi) of global rank `rank` under the group's lead share (what pk_load_dev slices and W slices must follow)"""
        lo, hi = C.c_uint64(), C.c_uint64()
        self._ck(self.lib.mi_group_wire_range(self.h, C.c_uint64(nb_wires), C.c_int(rank), C.byref(lo), C.byref(hi)))
        return int(lo.value), int(hi.value)

    def set_sharded_compute_h(self, on=True):
        self._ck(self.lib.mi_group_set_sharded_compute_h(self.h, C.c_uint32(1 if on else 0)))

    def compute_h_sharded_dev(self, log_n, a_ptrs, b_ptrs, c_ptrs, n_constraints, h_ptrs):
        """computeH over the ranks: per LOCAL rank device pointers to its rows of a, b (c_ptrs None: c = a o b) and to its M-element h slice"""
        arr = lambda ps: (C.c_void_p * len(ps))(*[C.c_void_p(int(p)) for p in ps])
        self._ck(self.lib.mi_compute_h_sharded_dev(self.h, C.c_uint32(log_n), arr(a_ptrs), arr(b_ptrs), None if c_ptrs is None else arr(c_ptrs),
                                                   C.c_size_t(n_constraints), arr(h_ptrs)))

    def exchange_selftest(self, nbytes=4096):
        self._ck(self.lib.mi_group_exchange_selftest(self.h, C.c_size_t(nbytes)))

    def pk_load(self, pk: dict):
        d, keep = _fill_pk_desc(pk); h = C.c_void_p()
        self._ck(self.lib.mi_pk_load_sharded(self.h, C.byref(d), C.byref(h)))
        return h

    def pk_load_dev(self, pk: dict, slices):
        """pk: header + masks of the WHOLE key (host); slices[i] = {name: (device pointer on local rank i's device, count)} for
        g1_a, g1_b, g1_k, g1_z, g2_b: that rank's slices (mi_pk_load_sharded_dev)"""
        descs = (PkDesc * len(slices))(); keep = []
        ia = np.ascontiguousarray(pk["infinity_a"], dtype=np.uint8); ib = np.ascontiguousarray(pk["infinity_b"], dtype=np.uint8)
        cw = pk.get("committed_wires")
        cw = np.ascontiguousarray(cw, dtype=np.uint32) if cw is not None and len(cw) else None
        keep += [ia, ib, cw]
        for d, sl in zip(descs, slices):
            d.log_n, d.nb_public, d.nb_wires = pk["log_n"], pk["nb_public"], pk["nb_wires"]
            for name in ("g1_a", "g1_b", "g1_k", "g1_z", "g2_b"):
                ptr, cnt = sl[name]
                setattr(d, name, int(ptr)); setattr(d, "n_" + name, int(cnt))
            for name, k in (("alpha1", 8), ("beta1", 8), ("delta1", 8), ("beta2", 16), ("delta2", 16)):
                setattr(d, name, (C.c_uint64 * k)(*[int(v) for v in _u64(pk[name]).reshape(-1)]))
            d.infinity_a, d.infinity_b = ia.ctypes.data, ib.ctypes.data
            if cw is not None:
                d.committed_wires, d.n_committed = cw.ctypes.data, cw.shape[0]
        h = C.c_void_p()
        self._ck(self.lib.mi_pk_load_sharded_dev(self.h, descs, C.byref(h)))
        return h

    def pk_free(self, spk):
        self._ck(self.lib.mi_pk_sharded_free(self.h, spk))

    def prove(self, spk, W, a, b, c, r, s, mode=0):
        """host inputs; a, b, c may be None in a process that does not hold rank 0"""
        out = np.zeros(32, np.uint64); st = Stats()
        W, r, s = _u64(W), _u64(r), _u64(s)
        a, b, c = (None if x is None else _u64(x) for x in (a, b, c))
        self._ck(self.lib.mi_groth16_prove_sharded(self.h, spk, _p(W), C.c_size_t(W.shape[0]), _p(a), _p(b), _p(c), C.c_size_t(0 if a is None else a.shape[0]),
                                                   _p(r), _p(s), C.c_uint32(mode), _p(out), C.byref(st)))
        return {"ar": out[:8].copy(), "bs": out[8:24].copy(), "krs": out[24:].copy(), "raw": out}, st.as_dict()

    def prove_dev(self, spk, W_ptrs, n_wires, a_ptr, b_ptr, c_ptr, n_constraints, r, s, mode=0):
        """inputs resident: W_ptrs[i] = wire range of local rank i on its device; a, b, c on rank 0's device (None elsewhere)"""
        out = np.zeros(32, np.uint64); st = Stats(); r, s = _u64(r), _u64(s)
        ww = (C.c_void_p * len(W_ptrs))(*[int(p) for p in W_ptrs])
        self._ck(self.lib.mi_groth16_prove_sharded_dev(self.h, spk, ww, C.c_size_t(n_wires), _p(a_ptr), _p(b_ptr), _p(c_ptr), C.c_size_t(n_constraints),
                                                       _p(r), _p(s), C.c_uint32(mode), _p(out), C.byref(st)))
        return {"ar": out[:8].copy(), "bs": out[8:24].copy(), "krs": out[24:].copy(), "raw": out}, st.as_dict()

    def prove_slices_dev(self, spk, W_ptrs, n_wires, a_ptrs, b_ptrs, c_ptrs, n_constraints, r, s, mode=0):
        """computeH over the ranks: a_ptrs[i] / b_ptrs[i] / c_ptrs[i] = local rank i's ROWS of a, b, c on its device (c_ptrs None: c = a o b)"""
        out = np.zeros(32, np.uint64); st = Stats(); r, s = _u64(r), _u64(s)
        arr = lambda ps: (C.c_void_p * len(ps))(*[C.c_void_p(int(p)) for p in ps])
        self._ck(self.lib.mi_groth16_prove_sharded_slices_dev(self.h, spk, arr(W_ptrs), C.c_size_t(n_wires), arr(a_ptrs), arr(b_ptrs), None if c_ptrs is None else arr(c_ptrs),
                                                              C.c_size_t(n_constraints), _p(r), _p(s), C.c_uint32(mode), _p(out), C.byref(st)))
        return {"ar": out[:8].copy(), "bs": out[8:24].copy(), "krs": out[24:].copy(), "raw": out}, st.as_dict()

    def msm_g1(self, pts, sc, flags=0, mode=0):
        out = np.zeros(12, np.uint64); pts, sc = _u64(pts), _u64(sc)
        self._ck(self.lib.mi_msm_g1_sharded(self.h, _p(pts), _p(sc), C.c_size_t(pts.shape[0]), C.c_uint32(flags), C.c_uint32(mode), _p(out)))
        return out

    def msm_g2(self, pts, sc, flags=0, mode=0):
        out = np.zeros(24, np.uint64); pts, sc = _u64(pts), _u64(sc)
        self._ck(self.lib.mi_msm_g2_sharded(self.h, _p(pts), _p(sc), C.c_size_t(pts.shape[0]), C.c_uint32(flags), C.c_uint32(mode), _p(out)))
        return out

    def msm_dev(self, pts_ptrs, sc_ptrs, n_local, n_total, flags=0, mode=0, g2=False):
        """pairs already resident: one device pointer / count per LOCAL rank"""
        k = len(n_local)
        pp = (C.c_void_p * k)(*[int(p) for p in pts_ptrs]); ss = (C.c_void_p * k)(*[int(p) for p in sc_ptrs]); nn = (C.c_size_t * k)(*n_local)
        out = np.zeros(24 if g2 else 12, np.uint64)
        f = self.lib.mi_msm_g2_sharded_dev if g2 else self.lib.mi_msm_g1_sharded_dev
        self._ck(f(self.h, pp, ss, nn, C.c_size_t(n_total), C.c_uint32(flags), C.c_uint32(mode), _p(out)))
        return out

    def close(self):
        if self.h:
            self.lib.mi_group_destroy(self.h)
            self.h = None


def proof_write(raw, commitments=None, pok=None):
    n = 0 if commitments is None else commitments.shape[0]
    buf = np.zeros(164 + 32 * n, np.uint8)
    ln = load().mi_proof_write(_p(_u64(raw)), _p(commitments), C.c_uint32(n), _p(pok), _p(buf))
    return bytes(buf[:ln])


def g1_compress(p):
    buf = np.zeros(32, np.uint8); load().mi_g1_compress(_p(_u64(p)), _p(buf)); return bytes(buf)


def g2_compress(p):
    buf = np.zeros(64, np.uint8); load().mi_g2_compress(_p(_u64(p)), _p(buf)); return bytes(buf)


def pedersen_fold(points, challenge):
    points = _u64(points); out = np.zeros(8, np.uint64)
    assert load().mi_pedersen_fold(_p(points), C.c_size_t(points.shape[0]), _p(_u64(challenge)), _p(out)) == 0
    return out


def g1_sum(parts):
    parts = _u64(parts); out = np.zeros(12, np.uint64)
    assert load().mi_g1_sum(_p(parts), C.c_size_t(parts.shape[0]), _p(out)) == 0
    return out


def g2_sum(parts):
    parts = _u64(parts); out = np.zeros(24, np.uint64)
    assert load().mi_g2_sum(_p(parts), C.c_size_t(parts.shape[0]), _p(out)) == 0
    return out
