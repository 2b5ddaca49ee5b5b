"""ORACLE (test infrastructure, never the product): restatement of the reference's HOST INGESTION of ProveKit artefacts -- SURVEY 8f N4 --
in plain Python, each function citing the reference lines it follows.  Only tests/ import this.

What it restates
  ark_decode_*        the arkworks canonical wire format as main.go:101,146 reads it through go-ark-serialize
                      (github.com/reilabs/go-ark-serialize, go.mod:10 -- third-party, NOT in /root/reference; restated from the
                      published ark-serialize rules, compress = false, validate = false, onto the Go struct shapes of main.go:15-39,74-76):
                          u64 / usize          8 bytes little-endian
                          [u8; 32]             32 raw bytes, no length                        (KeccakDigest, main.go:15-17)
                          Fp256                4 x u64 little-endian limbs, limb 0 first      (main.go:19-21; canonical, not Montgomery)
                          Vec<T> / []T         u64 length, then the elements
                          struct               the fields in declaration order
  reverse             utilities/utilities.go:58-65
  prefix_decode_path  utilities/utilities.go:67-78
  limbs_to_bigint_mod typeConverters/typeConverters.go:26-44
  parse_paths_object  mt.go:229-304 (the decoded per-leaf authentication paths, root end LAST after Reverse; leaf sibling hashes;
                      leaf indexes; leaves reduced mod r)
  matrix_cells        mt.go:358-401 (CSR with interned values -> (row, column, value) cells)
  parse_config        main.go:41-58,115 (encoding/json into Config)
and, for the fixtures only, the inverse direction (ark_encode_*, prefix_encode_paths, a toy Merkle tree).

PARITY UNPINNED like the rest of the oracle: the reference holds no ProveKit artefact and no test for this path, and go-ark-serialize
is absent, so nothing reference-held pins the wire format; first contact with a real `proof` file is the check."""
import hashlib
import json
import struct

R_MOD = 21888242871839275222246405745257275088548364400416034343698204186575808495617   # typeConverters/typeConverters.go:28


# ---------------------------------------------------------------- arkworks canonical reader / writer
class Reader:
    def __init__(self, buf: bytes):
        self.b, self.i = buf, 0

    def take(self, n):
        if self.i + n > len(self.b):
            raise ValueError("truncated input")
        out = self.b[self.i:self.i + n]
        self.i += n
        return out

    def u64(self):
        return struct.unpack("<Q", self.take(8))[0]

    def vec(self, elem):
        n = self.u64()
        if n > len(self.b) - self.i:   # every element takes at least one byte: a longer vector cannot be there
            raise ValueError("vector length exceeds the input")
        return [elem() for _ in range(n)]

    def digest(self):
        return self.take(32)

    def fp256(self):
        return list(struct.unpack("<4Q", self.take(32)))


def ark_decode_multipath(r: Reader):
    """MultiPath[KeccakDigest], main.go:23-28"""
    return {"leaf_sibling_hashes": r.vec(r.digest), "auth_paths_prefix_lengths": r.vec(r.u64),
            "auth_paths_suffixes": r.vec(lambda: r.vec(r.digest)), "leaf_indexes": r.vec(r.u64)}


def ark_decode_proof_element(r: Reader):
    """ProofElement, main.go:30-33"""
    return {"a": ark_decode_multipath(r), "b": r.vec(lambda: r.vec(r.fp256))}


def ark_decode_proof_object(buf: bytes):
    """ProofObject, main.go:35-39, as CanonicalDeserializeWithMode(proofFile, &proof, false, false) fills it (main.go:101)"""
    r = Reader(buf)
    out = {"round0_merkle_paths": r.vec(lambda: ark_decode_proof_element(r)), "merkle_paths": r.vec(lambda: ark_decode_proof_element(r)),
           "statement_values_at_random_point": r.vec(r.fp256)}
    return out, r.i


def ark_decode_interner(buf: bytes):
    """Interner{Values []Fp256}, main.go:74-76, read at main.go:146 from the hex string of r1cs.json"""
    r = Reader(buf)
    return r.vec(r.fp256), r.i


def _u64(x):
    return struct.pack("<Q", x)


def _vec(items, enc):
    return _u64(len(items)) + b"".join(enc(x) for x in items)


def ark_encode_multipath(m):
    return (_vec(m["leaf_sibling_hashes"], bytes) + _vec(m["auth_paths_prefix_lengths"], _u64) +
            _vec(m["auth_paths_suffixes"], lambda s: _vec(s, bytes)) + _vec(m["leaf_indexes"], _u64))


def ark_encode_fp256(limbs):
    return struct.pack("<4Q", *limbs)


def ark_encode_proof_element(e):
    return ark_encode_multipath(e["a"]) + _vec(e["b"], lambda leaf: _vec(leaf, ark_encode_fp256))


def ark_encode_proof_object(p):
    return (_vec(p["round0_merkle_paths"], ark_encode_proof_element) + _vec(p["merkle_paths"], ark_encode_proof_element) +
            _vec(p["statement_values_at_random_point"], ark_encode_fp256))


def ark_encode_interner(values):
    return _vec(values, ark_encode_fp256)


# ---------------------------------------------------------------- the reference's helpers
def reverse(s):
    """utilities/utilities.go:58-65"""
    return list(s[::-1])


def prefix_decode_path(prev_path, prefix_len, suffix):
    """utilities/utilities.go:67-78: prefix_len == 0 -> the suffix alone; else the first prefix_len nodes of the previous path, then the suffix"""
    if prefix_len == 0:
        return list(suffix)
    if prefix_len > len(prev_path):
        raise ValueError("prefix longer than the previous path")   # (Go would panic on the slice bounds)
    return list(prev_path[:prefix_len]) + list(suffix)


def limbs_to_bigint_mod(limbs):
    """typeConverters/typeConverters.go:26-44: limbs[0] + limbs[1] 2^64 + limbs[2] 2^128 + limbs[3] 2^192, reduced mod r"""
    return (limbs[0] + (limbs[1] << 64) + (limbs[2] << 128) + (limbs[3] << 192)) % R_MOD


def parse_paths_object(proof_elements):
    """mt.go:229-304 (the valued half; the "container" arrays it also builds are zero-filled shapes): per proof element
         auth_paths[j]          the authentication path of leaf j, leaf end first (Reverse of the decoded root-first path), 32-byte nodes
         leaf_sibling_hashes[j] 32 bytes
         leaf_indexes[j]        u64
         leaves[j]              the leaf's field elements, each LimbsToBigIntMod of its four limbs
       The tree height is len(AuthPathsSuffixes[0]) (mt.go:243): the first path is stored whole, every later one as a prefix length into
       the previous DECODED path plus its own suffix (mt.go:272-281)."""
    out = []
    for el in proof_elements:
        a = el["a"]
        n = len(a["leaf_indexes"])
        if n == 0 or not a["auth_paths_suffixes"]:
            raise ValueError("a proof element without leaves")   # (Go would panic at AuthPathsSuffixes[0])
        height = len(a["auth_paths_suffixes"][0])
        prev = list(a["auth_paths_suffixes"][0])
        paths = [reverse(prev)]
        for j in range(1, n):
            prev = prefix_decode_path(prev, a["auth_paths_prefix_lengths"][j], a["auth_paths_suffixes"][j])
            if len(prev) != height:
                raise ValueError("decoded path does not have the tree's height")   # (Go would index past the path at mt.go:279)
            paths.append(reverse(prev))
        out.append({"tree_height": height, "auth_paths": paths, "leaf_sibling_hashes": [a["leaf_sibling_hashes"][z] for z in range(n)],
                    "leaf_indexes": [a["leaf_indexes"][z] for z in range(n)],
                    "leaves": [[limbs_to_bigint_mod(x) for x in el["b"][z]] for z in range(n)]})
    return out


def matrix_cells(row_indices, col_indices, values, interner_values):
    """mt.go:358-401: CSR (row start offsets, column per entry, interner index per entry) -> cells (row, column, value mod r) in entry order;
    row i owns entries [row_indices[i], row_indices[i + 1] - 1], the last row runs to the end"""
    cells = [None] * len(values)
    for i in range(len(row_indices)):
        end = len(values) - 1
        if i < len(row_indices) - 1:
            end = row_indices[i + 1] - 1
        for j in range(row_indices[i], end + 1):
            cells[j] = (i, col_indices[j], limbs_to_bigint_mod(interner_values[values[j]]))
    return cells


CONFIG_INT_FIELDS = ("log_num_constraints", "n_rounds", "n_vars", "final_queries", "final_pow_bits", "final_folding_pow_bits", "rate", "transcript_len")
CONFIG_INT_LIST_FIELDS = ("folding_factor", "ood_samples", "num_queries", "pow_bits")


def parse_config(text: str):
    """main.go:41-58 + json.Unmarshal at main.go:115: missing keys keep Go's zero values; `transcript` is a []byte: encoding/json takes a
    JSON array of numbers 0..255 (what serde_json writes for a Vec<u8>, i.e. what ProveKit's params file holds) element by element, and
    a base64 string as a whole"""
    import base64
    j = json.loads(text)
    cfg = {k: int(j.get(k, 0)) for k in CONFIG_INT_FIELDS}
    for k in CONFIG_INT_LIST_FIELDS:
        cfg[k] = [int(x) for x in (j.get(k) or [])]
    cfg["domain_generator"] = j.get("domain_generator", "")
    cfg["io_pattern"] = j.get("io_pattern", "")
    t = j.get("transcript")
    if isinstance(t, str):
        cfg["transcript"] = base64.b64decode(t)
    elif t is None:
        cfg["transcript"] = b""
    else:
        if any((not isinstance(x, int)) or x < 0 or x > 255 for x in t):
            raise ValueError("transcript: not a byte")
        cfg["transcript"] = bytes(t)
    cfg["statement_evaluations"] = list(j.get("statement_evaluations") or [])
    return cfg


# ---------------------------------------------------------------- fixtures: a toy tree and the prover-side encoding
def toy_tree(height, seed):
    """a full binary tree of `height` levels below the root over 2^height leaves; node hash = sha256 (a stand-in: nothing here verifies
    hashes, only the path bookkeeping).  levels[0] = leaves ... levels[height] = [root]"""
    levels = [[hashlib.sha256(f"{seed}:{i}".encode()).digest() for i in range(1 << height)]]
    while len(levels[-1]) > 1:
        prev = levels[-1]
        levels.append([hashlib.sha256(prev[2 * i] + prev[2 * i + 1]).digest() for i in range(len(prev) // 2)])
    return levels


def root_first_path(levels, leaf):
    """the inner siblings of `leaf` from the level below the root down to the level above the leaves, then nothing for the leaf level
    (the leaf's own sibling travels separately as LeafSiblingHashes): height - 1 nodes... the reference stores `tree height` nodes per
    path (mt.go:243) -- the toy keeps one node per level 1..height, root end first"""
    height = len(levels) - 1
    path = []
    for lvl in range(height, 0, -1):   # node of `leaf` at level lvl - 1 is leaf >> (lvl - 1); its sibling there
        idx = (leaf >> (lvl - 1)) ^ 1
        path.append(levels[lvl - 1][idx])
    return path


def prefix_encode_paths(paths):
    """what the prover side does before serialising (the inverse of mt.go:272-281): path 0 whole, path j as (shared prefix with path j - 1, rest)"""
    prefix_lengths, suffixes = [0], [list(paths[0])]
    for j in range(1, len(paths)):
        k = 0
        while k < len(paths[j]) and paths[j][k] == paths[j - 1][k]:
            k += 1
        prefix_lengths.append(k)
        suffixes.append(list(paths[j][k:]))
    return prefix_lengths, suffixes
