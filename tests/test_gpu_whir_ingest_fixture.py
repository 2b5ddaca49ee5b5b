"""GPU-box record of SURVEY 8f N4: the ProveKit ingestion entry points (mi_whir_*, csrc/whir_ingest.hip) of the SHIPPED library decode the
committed fixture tests/golden/whir_proof_small.bin and the params file tests/golden/whir_params_small.json, and must reproduce the
committed expectations (whir_proof_small_full.json: per element the tree height, leaf indexes, leaf lengths and sha256 digests of the
authentication paths, the leaf sibling hashes and the leaf values mod r).  DATA ONLY: the Python restatement that wrote those files
(oracle/whir_ingest.py through oracle/gen_whir_fixture.py) stays in the build container; the same comparison against the restatement
itself runs there (tests/test_whir_ingest.py).  The entry points are host code: marked gpu so that the driver's GPU-box run loads them
from the in-tree libmi355x_groth16.so it records."""
import ctypes as C
import hashlib
import json
import os
import numpy as np
import pytest
from gpu_common import load_binding, ROOT

pytestmark = pytest.mark.gpu
GOLD = os.path.join(ROOT, "tests", "golden")
R_MOD = 21888242871839275222246405745257275088548364400416034343698204186575808495617


class Shape(C.Structure):
    _fields_ = [("n_leaves", C.c_uint64), ("tree_height", C.c_uint64), ("total_leaf_values", C.c_uint64)]


class Config(C.Structure):
    _fields_ = ([(n, C.c_int64) for n in ("log_num_constraints", "n_rounds", "n_vars", "final_queries", "final_pow_bits", "final_folding_pow_bits", "rate", "transcript_len")] +
                [(n, C.c_int64 * 64) for n in ("folding_factor", "ood_samples", "num_queries", "pow_bits")] +
                [(n, C.c_uint32) for n in ("n_folding_factor", "n_ood_samples", "n_num_queries", "n_pow_bits")] +
                [("domain_generator", C.c_uint64 * 4), ("io_pattern", C.c_void_p), ("io_pattern_len", C.c_size_t), ("transcript", C.c_void_p), ("n_transcript", C.c_size_t),
                 ("statement_evaluations", C.c_void_p), ("n_statement_evaluations", C.c_size_t), ("store", C.c_void_p)])


@pytest.fixture(scope="module")
def lib():
    L = load_binding().load()
    L.mi_whir_proof_elements.restype = C.c_uint64
    L.mi_whir_proof_statement_values.restype = C.c_uint64
    L.mi_whir_proof_free.restype = None
    L.mi_whir_limbs_to_fr.restype = None
    L.mi_whir_config_free.restype = None
    return L


def test_proof_object_fixture_decodes_to_the_committed_expectation(lib):
    buf = open(os.path.join(GOLD, "whir_proof_small.bin"), "rb").read()
    exp = json.load(open(os.path.join(GOLD, "whir_proof_small_full.json")))
    assert hashlib.sha256(buf).hexdigest() == exp["sha256"]
    h = C.c_void_p(); used = C.c_size_t()
    assert lib.mi_whir_proof_decode(buf, C.c_size_t(len(buf)), C.byref(h), C.byref(used)) == 0 and used.value == len(buf)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    try:
        for which, key in ((0, "round0_merkle_paths"), (1, "merkle_paths")):
            assert lib.mi_whir_proof_elements(h, which) == len(exp[key])
            for i, e in enumerate(exp[key]):
                sh = Shape()
                assert lib.mi_whir_element_shape(h, which, C.c_uint64(i), C.byref(sh)) == 0
                assert sh.tree_height == e["tree_height"] and sh.n_leaves == len(e["leaf_indexes"]) and sh.total_leaf_values == sum(e["leaf_lengths"])
                n, ht, tot = sh.n_leaves, sh.tree_height, sh.total_leaf_values
                paths = np.zeros((n, ht, 32), np.uint8); sib = np.zeros((n, 32), np.uint8); idx = np.zeros(n, np.uint64); lens = np.zeros(n, np.uint64)
                leaves = np.zeros((tot, 4), np.uint64)
                assert lib.mi_whir_parse_paths(h, which, C.c_uint64(i), vp(paths), vp(sib), vp(idx), vp(lens), vp(leaves)) == 0
                assert list(map(int, idx)) == e["leaf_indexes"] and list(map(int, lens)) == e["leaf_lengths"]
                assert hashlib.sha256(paths.tobytes()).hexdigest() == e["auth_paths_sha256"]
                assert hashlib.sha256(sib.tobytes()).hexdigest() == e["leaf_sibling_hashes_sha256"]
                assert hashlib.sha256(leaves.astype("<u8").tobytes()).hexdigest() == e["leaves_mod_r_sha256"]   # 4 x u64 little-endian = 32 bytes little-endian
        n = lib.mi_whir_proof_statement_values(h, None)
        st = np.zeros((n, 4), np.uint64)
        lib.mi_whir_proof_statement_values(h, vp(st))
        assert [list(map(int, r)) for r in st] == exp["statement_values_limbs"]
        for row, want in zip(st, exp["statement_values_mod_r"]):   # LimbsToBigIntMod, typeConverters/typeConverters.go:26-44
            out = np.zeros(4, np.uint64)
            lib.mi_whir_limbs_to_fr(vp(np.ascontiguousarray(row)), vp(out))
            got = sum(int(out[q]) << (64 * q) for q in range(4))
            assert str(got) == want and got == sum(int(row[q]) << (64 * q) for q in range(4)) % R_MOD
    finally:
        lib.mi_whir_proof_free(h)
    # a truncated stream and a corrupted length word are refused by the shipped library too
    bad = bytearray(buf); bad[8:16] = (1 << 40).to_bytes(8, "little")
    for raw in (buf[:-1], bytes(bad), b""):
        h2 = C.c_void_p()
        rc = lib.mi_whir_proof_decode(raw, C.c_size_t(len(raw)), C.byref(h2), None)
        if raw == b"":
            assert rc != 0 or lib.mi_whir_proof_elements(h2, 0) == 0   # (an empty stream has no length word: refused)
        else:
            assert rc != 0 and not h2


def test_params_fixture_parses_to_the_committed_expectation(lib):
    fx = json.load(open(os.path.join(GOLD, "whir_params_small.json"), encoding="utf-8"))
    raw = fx["text"].encode()
    exp = fx["expect"]
    h = C.POINTER(Config)()
    assert lib.mi_whir_config_parse(raw, C.c_size_t(len(raw)), C.byref(h)) == 0
    c = h.contents
    for k in ("log_num_constraints", "n_rounds", "n_vars", "final_queries", "final_pow_bits", "final_folding_pow_bits", "rate", "transcript_len"):
        assert getattr(c, k) == exp[k], k
    for k in ("folding_factor", "ood_samples", "num_queries", "pow_bits"):
        assert list(getattr(c, k))[:getattr(c, "n_" + k)] == exp[k], k
    assert sum(int(c.domain_generator[q]) << (64 * q) for q in range(4)) == int(exp["domain_generator"])
    assert C.string_at(c.io_pattern, c.io_pattern_len) == exp["io_pattern"].encode()
    assert list(C.string_at(c.transcript, c.n_transcript)) == exp["transcript"]
    ev = (C.c_uint64 * (4 * c.n_statement_evaluations)).from_address(c.statement_evaluations)
    assert [str(sum(int(ev[4 * i + q]) << (64 * q) for q in range(4))) for i in range(c.n_statement_evaluations)] == exp["statement_evaluations"]
    lib.mi_whir_config_free(h)
