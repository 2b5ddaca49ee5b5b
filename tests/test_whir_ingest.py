"""CPU: the ProveKit ingestion entry points (mi_whir_*, csrc/whir_ingest.hip -- pure host code of the C-ABI library) against
oracle/whir_ingest.py, the Python restatement of the reference's own loops (main.go:15-58,101,115,146; mt.go:229-304,358-401;
utilities/utilities.go:58-78; typeConverters/typeConverters.go:26-44), on seeded synthetic proof objects and on the committed fixture
tests/golden/whir_proof_small.bin (+ .json: what the oracle decodes from it).  Synthetic because the reference holds no ProveKit
artefact: the arkworks wire format is restated, not pinned (oracle/whir_ingest.py header)."""
import ctypes as C
import hashlib
import json
import os
import random
import struct
import numpy as np
import pytest
import whir_ingest as W
from gpu_common import load_binding, ROOT

GOLD = os.path.join(ROOT, "tests", "golden")


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


def synth_element(rng, height, n_leaves, leaf_len, seed):
    """one ProofElement the way the prover side would write it: sorted distinct leaf indexes of a toy tree, root-first paths,
    prefix-compressed against the previous path"""
    levels = W.toy_tree(height, seed)
    idx = sorted(rng.sample(range(1 << height), n_leaves))
    paths = [W.root_first_path(levels, i) for i in idx]
    pl, suf = W.prefix_encode_paths(paths)
    leaves = [[[rng.getrandbits(64) for _ in range(4)] for _ in range(leaf_len)] for _ in idx]   # raw limbs, some above r on purpose
    return {"a": {"leaf_sibling_hashes": [levels[0][i ^ 1] for i in idx], "auth_paths_prefix_lengths": pl, "auth_paths_suffixes": suf, "leaf_indexes": idx},
            "b": leaves}, paths


def synth_proof(seed, shapes0, shapes1, n_stmt=3):
    rng = random.Random(seed)
    mk = lambda shapes, tag: [synth_element(rng, h, n, ll, f"{seed}:{tag}:{k}") for k, (h, n, ll) in enumerate(shapes)]
    e0, e1 = mk(shapes0, 0), mk(shapes1, 1)
    proof = {"round0_merkle_paths": [e for e, _ in e0], "merkle_paths": [e for e, _ in e1],
             "statement_values_at_random_point": [[rng.getrandbits(64) for _ in range(4)] for _ in range(n_stmt)]}
    return proof, [p for _, p in e0], [p for _, p in e1]


def decode_with_lib(lib, buf):
    h = C.c_void_p(); used = C.c_size_t()
    rc = lib.mi_whir_proof_decode(buf, C.c_size_t(len(buf)), C.byref(h), C.byref(used))
    return rc, h, used.value


def parse_with_lib(lib, h, which, i):
    sh = Shape()
    assert lib.mi_whir_element_shape(h, which, C.c_uint64(i), C.byref(sh)) == 0
    n, ht, tot = sh.n_leaves, sh.tree_height, sh.total_leaf_values
    paths = np.zeros((n, ht, 32), np.uint8); sib = np.zeros((n, 32), np.uint8); idx = np.zeros(n, np.uint64); lens = np.zeros(n, np.uint64)
    leaves = np.zeros((tot, 4), np.uint64)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    rc = lib.mi_whir_parse_paths(h, which, C.c_uint64(i), vp(paths), vp(sib), vp(idx), vp(lens), vp(leaves))
    return rc, paths, sib, idx, lens, leaves


def check_against_oracle(lib, buf):
    want, used_want = W.ark_decode_proof_object(buf)
    rc, h, used = decode_with_lib(lib, buf)
    assert rc == 0 and used == used_want == len(buf)
    try:
        for which, key in ((0, "round0_merkle_paths"), (1, "merkle_paths")):
            assert lib.mi_whir_proof_elements(h, which) == len(want[key])
            parsed = W.parse_paths_object(want[key])
            for i, pw in enumerate(parsed):
                rc, paths, sib, idx, lens, leaves = parse_with_lib(lib, h, which, i)
                assert rc == 0
                assert paths.shape[1] == pw["tree_height"]
                assert [[bytes(paths[j, z]) for z in range(paths.shape[1])] for j in range(paths.shape[0])] == pw["auth_paths"]
                assert [bytes(x) for x in sib] == pw["leaf_sibling_hashes"] and list(map(int, idx)) == pw["leaf_indexes"]
                assert list(map(int, lens)) == [len(x) for x in pw["leaves"]]
                flat = [v for leaf in pw["leaves"] for v in leaf]
                assert [sum(int(leaves[k, q]) << (64 * q) for q in range(4)) for k in range(leaves.shape[0])] == flat
        n = lib.mi_whir_proof_statement_values(h, None)
        st = np.zeros((n, 4), np.uint64)
        lib.mi_whir_proof_statement_values(h, st.ctypes.data_as(C.c_void_p))
        assert [list(map(int, r)) for r in st] == want["statement_values_at_random_point"]
    finally:
        lib.mi_whir_proof_free(h)
    return want


def test_synthetic_proof_objects_decode_like_the_oracle_and_recover_the_true_paths(lib):
    for seed, s0, s1 in ((1, [(4, 3, 2)], [(3, 2, 1)]), (2, [(10, 40, 16)], [(9, 33, 16), (8, 29, 4), (7, 1, 4)]), (3, [], [(5, 32, 1)]), (4, [(6, 5, 0)], [])):
        proof, paths0, paths1 = synth_proof(seed, s0, s1)
        buf = W.ark_encode_proof_object(proof)
        want = check_against_oracle(lib, buf)
        assert want == proof   # the oracle's reader inverts its writer
        # ... and the decoded paths ARE the tree's paths, leaf end first (what Reverse leaves, mt.go:269,277)
        for key, true_paths in (("round0_merkle_paths", paths0), ("merkle_paths", paths1)):
            for el, tp in zip(W.parse_paths_object(want[key]), true_paths):
                assert el["auth_paths"] == [p[::-1] for p in tp]


def fixture_expectation(dec):
    return {"elements": [len(dec["round0_merkle_paths"]), len(dec["merkle_paths"])],
            "first_round_first_path_leaf_end": W.parse_paths_object(dec["round0_merkle_paths"])[0]["auth_paths"][0][0].hex(),
            "merkle_last_leaf_index": W.parse_paths_object(dec["merkle_paths"])[-1]["leaf_indexes"][-1],
            "first_leaf_value_mod_r": str(W.parse_paths_object(dec["merkle_paths"])[0]["leaves"][0][0]),
            "statement_values": [str(W.limbs_to_bigint_mod(x)) for x in dec["statement_values_at_random_point"]]}


def test_committed_fixture(lib):
    buf = open(os.path.join(GOLD, "whir_proof_small.bin"), "rb").read()
    exp = json.load(open(os.path.join(GOLD, "whir_proof_small.json")))
    assert hashlib.sha256(buf).hexdigest() == exp["sha256"]
    want = check_against_oracle(lib, buf)
    assert fixture_expectation(want) == exp["expect"]


def test_malformed_streams_are_refused_not_trusted(lib):
    proof, _, _ = synth_proof(9, [(5, 4, 2)], [(4, 3, 2)])
    buf = W.ark_encode_proof_object(proof)
    for cut in (0, 7, 8, 40, len(buf) // 2, len(buf) - 1):
        rc, h, _ = decode_with_lib(lib, buf[:cut])
        assert rc != 0 and not h.value
        with pytest.raises(ValueError):
            W.ark_decode_proof_object(buf[:cut])
    huge = struct.pack("<Q", 1 << 60) + buf[8:]   # a vector length the input cannot hold: refused before anything is reserved
    assert decode_with_lib(lib, huge)[0] != 0
    # a prefix length longer than the previous path / a decoded path of the wrong height: MI_EINVAL where Go would panic
    for mutate in (lambda a: a["auth_paths_prefix_lengths"].__setitem__(1, 99), lambda a: a["auth_paths_suffixes"][1].append(b"\0" * 32)):
        p2, _, _ = synth_proof(9, [(5, 4, 2)], [(4, 3, 2)])
        mutate(p2["round0_merkle_paths"][0]["a"])
        b2 = W.ark_encode_proof_object(p2)
        rc, h, _ = decode_with_lib(lib, b2)
        assert rc == 0
        assert parse_with_lib(lib, h, 0, 0)[0] != 0
        lib.mi_whir_proof_free(h)
        with pytest.raises(ValueError):
            W.parse_paths_object(W.ark_decode_proof_object(b2)[0]["round0_merkle_paths"])


def test_reverse_prefix_decode_and_limbs_helpers(lib):
    rng = random.Random(5)
    items = [bytes(rng.getrandbits(8) for _ in range(32)) for _ in range(7)]
    out = C.create_string_buffer(32 * 7)
    assert lib.mi_whir_reverse(b"".join(items), C.c_size_t(7), C.c_size_t(32), out) == 0
    assert [out.raw[32 * i:32 * i + 32] for i in range(7)] == W.reverse(items)
    n = C.c_size_t()
    for pl, suf in ((0, items[:3]), (2, items[3:]), (7, []), (0, [])):
        out = C.create_string_buffer(32 * 16)
        assert lib.mi_whir_prefix_decode_path(b"".join(items), C.c_size_t(7), C.c_uint64(pl), b"".join(suf), C.c_size_t(len(suf)), C.c_size_t(32), out, C.byref(n)) == 0
        assert [out.raw[32 * i:32 * i + 32] for i in range(n.value)] == W.prefix_decode_path(items, pl, suf)
    assert lib.mi_whir_prefix_decode_path(b"".join(items), C.c_size_t(7), C.c_uint64(8), b"", C.c_size_t(0), C.c_size_t(32), out, C.byref(n)) != 0
    for limbs in ([[0, 0, 0, 0], [1, 2, 3, 4], [2 ** 64 - 1] * 4, [0x43e1f593f0000001, 0x2833e84879b97091, 0xb85045b68181585d, 0x30644e72e131a029],
                   [0x43e1f593f0000000, 0x2833e84879b97091, 0xb85045b68181585d, 0x30644e72e131a029]] + [[rng.getrandbits(64) for _ in range(4)] for _ in range(50)]):
        o = (C.c_uint64 * 4)()
        lib.mi_whir_limbs_to_fr((C.c_uint64 * 4)(*limbs), o)
        assert sum(int(o[q]) << (64 * q) for q in range(4)) == W.limbs_to_bigint_mod(limbs)


def test_interner_and_matrix_cells(lib):
    rng = random.Random(6)
    interner = [[rng.getrandbits(64) for _ in range(4)] for _ in range(11)]
    buf = W.ark_encode_interner(interner)
    n = C.c_uint64(); used = C.c_size_t()
    assert lib.mi_whir_interner_decode(buf, C.c_size_t(len(buf)), None, C.byref(n), C.byref(used)) == 0 and n.value == 11 and used.value == len(buf)
    vals = np.zeros((11, 4), np.uint64)
    assert lib.mi_whir_interner_decode(buf, C.c_size_t(len(buf)), vals.ctypes.data_as(C.c_void_p), C.byref(n), None) == 0
    assert [list(map(int, r)) for r in vals] == interner == W.ark_decode_interner(buf)[0]
    # a CSR matrix with an empty row in the middle and at the end
    row_indices = [0, 2, 2, 5, 7, 7]
    nnz = 7
    cols = [rng.randrange(50) for _ in range(nnz)]; vidx = [rng.randrange(11) for _ in range(nnz)]
    want = W.matrix_cells(row_indices, cols, vidx, interner)
    u = lambda a: np.array(a, dtype=np.uint64)
    ri, ci, vi = u(row_indices), u(cols), u(vidx)
    ro, co, vo = np.zeros(nnz, np.uint64), np.zeros(nnz, np.uint64), np.zeros((nnz, 4), np.uint64)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    assert lib.mi_whir_matrix_cells(vp(ri), C.c_size_t(len(row_indices)), vp(ci), vp(vi), C.c_size_t(nnz), vp(vals), C.c_size_t(11), vp(ro), vp(co), vp(vo)) == 0
    got = [(int(ro[j]), int(co[j]), sum(int(vo[j, q]) << (64 * q) for q in range(4))) for j in range(nnz)]
    assert got == want
    vi[3] = 11   # an interner index out of range: Go panics, the library refuses
    assert lib.mi_whir_matrix_cells(vp(ri), C.c_size_t(len(row_indices)), vp(ci), vp(vi), C.c_size_t(nnz), vp(vals), C.c_size_t(11), vp(ro), vp(co), vp(vo)) != 0


CONFIG_JSON = json.dumps({
    "log_num_constraints": 17, "n_rounds": 4, "n_vars": 20, "folding_factor": [4, 4, 4, 4], "ood_samples": [2, 2, 1, 1], "num_queries": [103, 46, 30, 23],
    "pow_bits": [18, 20, 22, 22], "final_queries": 18, "final_pow_bits": 21, "final_folding_pow_bits": 0,
    "domain_generator": "19103219067921713944291392827692070036145651957329286315305642004821462161904", "rate": 1,
    "io_pattern": chr(0x1F32A) + chr(0xFE0F) + ' A32merkle_digest S47initial_"combination\\randomness' + chr(0xE9), "transcript": [0, 1, 2, 254, 255, 17], "transcript_len": 6,
    "statement_evaluations": ["0", "21888242871839275222246405745257275088548364400416034343698204186575808495616", "12345678901234567890123456789"],
    "an_unknown_key": {"nested": [1, 2, {"x": "y"}], "z": None}, "another": -3.5e2})


def test_config_json_like_encoding_json(lib):
    want = W.parse_config(CONFIG_JSON)
    for text in (CONFIG_JSON, json.dumps(json.loads(CONFIG_JSON), ensure_ascii=False)):   # \\uXXXX escapes (surrogate pairs) and raw UTF-8
        raw = text.encode()
        h = C.POINTER(Config)()
        assert lib.mi_whir_config_parse(raw, C.c_size_t(len(raw)), C.byref(h)) == 0
        c = h.contents
        for k in W.CONFIG_INT_FIELDS:
            assert getattr(c, k) == want[k]
        for k in W.CONFIG_INT_LIST_FIELDS:
            assert list(getattr(c, k))[:getattr(c, "n_" + k)] == want[k]
        assert sum(int(c.domain_generator[q]) << (64 * q) for q in range(4)) == int(want["domain_generator"])
        assert C.string_at(c.io_pattern, c.io_pattern_len) == want["io_pattern"].encode()
        assert C.string_at(c.transcript, c.n_transcript) == want["transcript"]
        ev = (C.c_uint64 * (4 * c.n_statement_evaluations)).from_address(c.statement_evaluations)
        assert [str(sum(int(ev[4 * i + q]) << (64 * q) for q in range(4))) for i in range(c.n_statement_evaluations)] == want["statement_evaluations"]
        lib.mi_whir_config_free(h)
    # base64 transcript, missing keys -> zero values
    h = C.POINTER(Config)()
    small = json.dumps({"n_vars": 3, "transcript": "AAEC/v8R"}).encode()
    assert lib.mi_whir_config_parse(small, C.c_size_t(len(small)), C.byref(h)) == 0
    assert h.contents.n_vars == 3 and h.contents.n_rounds == 0
    assert C.string_at(h.contents.transcript, h.contents.n_transcript) == bytes([0, 1, 2, 254, 255, 17]) == W.parse_config(small.decode())["transcript"]
    lib.mi_whir_config_free(h)
    for bad in (b"", b"[1]", b'{"n_vars": "3"}', b'{"n_vars": 3', b'{"transcript": [256]}', b'{"domain_generator": "12x"}', b'{"folding_factor": [1, 2.5]}'):
        assert lib.mi_whir_config_parse(bad, C.c_size_t(len(bad)), C.byref(h)) != 0
