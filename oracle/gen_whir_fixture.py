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

# The full expectation, element by element (what the GPU box's test compares the shipped library with -- data only, no oracle there):
# per element the tree height, the leaf indexes and leaf lengths, and sha256 digests of the authentication paths (leaf end first, as
# ParsePathsObject leaves them, mt.go:269,277), of the leaf sibling hashes and of the leaf values reduced mod r (32 bytes little-endian each).
def element_digest(pw):
    h = lambda chunks: hashlib.sha256(b"".join(chunks)).hexdigest()
    return {"tree_height": pw["tree_height"], "leaf_indexes": pw["leaf_indexes"], "leaf_lengths": [len(x) for x in pw["leaves"]],
            "auth_paths_sha256": h([d for path in pw["auth_paths"] for d in path]), "leaf_sibling_hashes_sha256": h(pw["leaf_sibling_hashes"]),
            "leaves_mod_r_sha256": h([int(v).to_bytes(32, "little") for leaf in pw["leaves"] for v in leaf])}
full = {"sha256": hashlib.sha256(buf).hexdigest(), "generator": "oracle/gen_whir_fixture.py",
        "round0_merkle_paths": [element_digest(pw) for pw in W.parse_paths_object(dec["round0_merkle_paths"])],
        "merkle_paths": [element_digest(pw) for pw in W.parse_paths_object(dec["merkle_paths"])],
        "statement_values_mod_r": [str(W.limbs_to_bigint_mod(x)) for x in dec["statement_values_at_random_point"]],
        "statement_values_limbs": dec["statement_values_at_random_point"]}
json.dump(full, open(os.path.join(ROOT, "tests", "golden", "whir_proof_small_full.json"), "w"), indent=1)
# a params file (Config, main.go:41-58) and what encoding/json leaves in the struct, as the oracle's reader restates it
from test_whir_ingest import CONFIG_JSON
cfg = W.parse_config(CONFIG_JSON)
cfg["transcript"] = list(cfg["transcript"])
json.dump({"generator": "oracle/gen_whir_fixture.py", "text": CONFIG_JSON, "expect": cfg}, open(os.path.join(ROOT, "tests", "golden", "whir_params_small.json"), "w"), indent=1)
print(len(buf), "bytes")
