"""CPU: the library's parsers of OUTSIDE bytes -- mi_whir_proof_decode (+ accessors, mi_whir_parse_paths), mi_whir_interner_decode,
mi_whir_config_parse, mi_whir_matrix_cells, mi_pk_raw_inspect -- built as plain C++ with -fsanitize=address,undefined
(gnark-whir_amd/Makefile `sanitize`: csrc/whir_ingest.hip + csrc/pk_raw_inspect.hip hold no HIP) and driven by the mutation driver
tests/cpp/parser_fuzz.cpp over valid seeds: every truncation, bit flips, corrupted length fields, random blobs, JSON token splices and
20000-deep nesting.  A sanitizer report or a broken contract (a section outside the input, a handle from a failed call) fails the test;
whether a mutant is accepted or refused does not.  Plus the encoding/json corner cases the reader must share with Go."""
import ctypes as C
import json
import os
import random
import subprocess
import sys
import pytest
import pyref as P
import cref
import pk_raw
import whir_ingest as W
from helpers import g1_pts
from gpu_common import load_binding, ROOT
from test_whir_ingest import CONFIG_JSON, Config

PKG = os.path.join(ROOT, "gnark-whir_amd")
GOLD = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def fuzz_bin():
    subprocess.check_call(["make", "-C", PKG, "-s", "sanitize"])
    return os.path.join(PKG, "build", "parser_fuzz_asan")


@pytest.fixture(scope="module")
def seeds(tmp_path_factory):
    d = tmp_path_factory.mktemp("parser_seeds")
    rng = random.Random(5)
    paths = {"proof": os.path.join(GOLD, "whir_proof_small.bin"), "config": str(d / "config.json"), "interner": str(d / "interner.bin"), "pk_raw": str(d / "pk_raw.bin")}
    open(paths["config"], "w", encoding="utf-8").write(CONFIG_JSON)
    open(paths["interner"], "wb").write(W.ark_encode_interner([[rng.getrandbits(64) for _ in range(4)] for _ in range(23)]))
    cs = P.ToyR1CS(60, 4, 31); td = P.ToyTrapdoor(31)
    pk, _, _ = P.toy_setup(cs, td)
    keys = [(g1_pts(cref.gen_g1(5 + k, 40 + k)), g1_pts(cref.gen_g1(5 + k, 50 + k))) for k in range(2)]
    open(paths["pk_raw"], "wb").write(pk_raw.write_pk_raw(pk, keys))
    return paths


@pytest.mark.parametrize("seed", [1, 2])
def test_mutated_inputs_never_trip_a_sanitizer(fuzz_bin, seeds, seed):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:allocator_may_return_null=1", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([fuzz_bin, seeds["proof"], seeds["config"], seeds["interner"], seeds["pk_raw"], "60000", str(seed)], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, f"rc {r.returncode}\n{r.stdout[-2000:]}\n{r.stderr[-6000:]}"
    counts = json.loads(r.stdout.strip().splitlines()[-1])
    for k, (acc, ref) in counts.items():
        assert acc > 0 and ref > 100, (k, acc, ref)   # the mutants reach both outcomes of every parser


def _parse(lib, raw):
    h = C.POINTER(Config)()
    rc = lib.mi_whir_config_parse(raw, C.c_size_t(len(raw)), C.byref(h))
    return rc, h


def test_config_reader_agrees_with_encoding_json_on_the_corner_cases():
    """what Go's json.Unmarshal into main.go's Config does with these texts (restated: no Go here): refuse / accept, and the field values"""
    lib = load_binding().load()
    lib.mi_whir_config_free.restype = None
    refuse = [b'{"folding_factor": nXYZ}', b'{"folding_factor": nul', b'{"folding_factor": n', b'{"n_vars": 9223372036854775808}', b'{"n_vars": -9223372036854775809}',
              b'{"n_vars": 01}', b'{"n_vars": 1.0}', b'{"n_vars": 1e2}', b'{"n_vars": -}', b'{"n_vars": tru}', b'{"x": tru}', b'{"x": nulll}', b'{"x": 1.}', b'{"x": .5}',
              b'{"io_pattern": "a\\xb"}', b'{"io_pattern": "a\nb"}', b'{"io_pattern": "\\u12"}', b'{"n_vars": 3} x', b'{"n_vars": 3}{', b'{"n_vars": 3,}', b'{,}', b'{"a" 1}',
              b'{"transcript": "AAE"}', b'{"transcript": "A=EC"}', b'{"transcript": 5}', b'{"transcript": true}', b'{"statement_evaluations": 7}',
              b'{"x": ' + b"[" * 10001 + b"]" * 10001 + b"}", b'{"x": ' + b"[" * 200000]
    for raw in refuse:
        rc, h = _parse(lib, raw)
        assert rc != 0 and not h, raw[:60]
    accept = {
        b'null': {},
        b' {"n_vars": null, "folding_factor": null, "io_pattern": null, "transcript": null, "statement_evaluations": null, "domain_generator": null} ': {"n_vars": 0, "n_folding_factor": 0},
        b'{"N_VARS": 7, "Folding_Factor": [1, null, 3]}': {"n_vars": 7, "n_folding_factor": 3},
        b'{"n_vars": 1, "n_vars": 2}': {"n_vars": 2},
        b'{"n_vars": -9223372036854775808, "rate": 9223372036854775807}': {"n_vars": -(1 << 63), "rate": (1 << 63) - 1},
        b'{"folding_factor": [1, 2], "folding_factor": null}': {"n_folding_factor": 2},
        b'{"x": ' + b"[" * 9999 + b"]" * 9999 + b', "n_vars": 4}': {"n_vars": 4},
        b'{"x": [true, false, null, -0, 1e9, 1.5E-3, {"y": "\\u00e9"}], "n_vars": 5}\r\n\t ': {"n_vars": 5},
    }
    for raw, want in accept.items():
        rc, h = _parse(lib, raw)
        assert rc == 0, raw[:60]
        for k, v in want.items():
            assert getattr(h.contents, k) == v, (raw[:60], k)
        lib.mi_whir_config_free(h)
    # an unpaired surrogate escape is U+FFFD, a pair is one code point; base64 with line breaks
    rc, h = _parse(lib, b'{"io_pattern": "a\\ud800b\\udc00c\\ud83c\\udf2a", "transcript": "AAEC\\r\\n/v8R"}')
    assert rc == 0
    assert C.string_at(h.contents.io_pattern, h.contents.io_pattern_len) == "a�b�c\U0001F32A".encode()
    assert C.string_at(h.contents.transcript, h.contents.n_transcript) == bytes([0, 1, 2, 254, 255, 17])
    lib.mi_whir_config_free(h)
