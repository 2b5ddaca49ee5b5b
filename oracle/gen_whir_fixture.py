"""Writes tests/golden/whir_proof_small.bin / .json: a seeded synthetic ProofObject in the arkworks canonical format as oracle/whir_ingest.py
restates it (two first-round elements, three later rounds), and what the oracle decodes from it.  Test infrastructure; run from the repo
root: python oracle/gen_whir_fixture.py"""
import hashlib
import json
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import whir_ingest as W
from test_whir_ingest import synth_proof, fixture_expectation

proof, _, _ = synth_proof(0x57484952, [(8, 12, 16), (8, 9, 16)], [(7, 10, 16), (6, 8, 4), (5, 6, 4)], n_stmt=3)
buf = W.ark_encode_proof_object(proof)
dec, _ = W.ark_decode_proof_object(buf)
open(os.path.join(ROOT, "tests", "golden", "whir_proof_small.bin"), "wb").write(buf)
json.dump({"sha256": hashlib.sha256(buf).hexdigest(), "bytes": len(buf), "generator": "oracle/gen_whir_fixture.py", "expect": fixture_expectation(dec)},
          open(os.path.join(ROOT, "tests", "golden", "whir_proof_small.json"), "w"), indent=1)
print(len(buf), "bytes")
