"""Wire-class census of the WHIR-verifier circuit: which share of the Groth16 witness is a bit, a byte, a <= 64-bit value or a full-width
field element -- the one input the benchmark's headline is hostage to.

    python tools/wire_census.py [--write profiles/r06_wire_census.txt] [--json]

Pure Python; it opens NO file of the reference at run time.  Every term below RESTATES a counting rule read off the reference's circuit
(/root/reference, file:line per term) under gnark's R1CS builder cost model (add / sub / multiply-by-constant are free;
var x var Mul, Select, And, Inverse cost one constraint and one internal wire; AssertIsEqual one constraint; ToBinary(n) n wires and
n + 1 constraints; hints only wires).  Third-party gadget internals that are not in the container (gnark-skyscraper's Compress, gnark-nimue's
Arthur, gnark std logderivlookup / uints tables) are NOT priced by recollection: each goes into an `unknown` bucket with the exact number of
CALLS the reference makes and an explicit lower / upper bound per call, stated where the bucket is defined.

What matters to the prover: the four wire-value MSMs (pk.G1.A, pk.G1.B, pk.G2.B, pk.G1.K) take the wire values W as scalars; a bit or a
byte is one non-zero Pippenger digit (or none), a full-width value 13-15.  (The h scalars of pk.G1.Z are uniform whatever the witness is;
computeH's transforms take the same time for any a, b, c.)

Value classes: `bit` {0,1} . `byte` any value below 2^16 (bytes, byte pairs, small indices, multiplicities: one digit) . `u64` up to 64
bits . `full` a full-width field element (challenges, hashes, products of challenges, inverses).
"""
import argparse
import json
import math
import os
import sys
from dataclasses import dataclass, field

CLASSES = ("bit", "byte", "u64", "full")


@dataclass
class Params:
    """What ProveKit's params file carries (main.go:41-58), for BASELINE configs[1]: 2^20-variable multilinear, folding factor 4, rate 1/2,
    128-bit security.  `log_m` (log_num_constraints of the inner R1CS), `nnz_per_row` (average non-zeros per row and matrix) and the soundness
    flavour / proof-of-work bits are NOT fixed by BASELINE.json: they are the census's scenario axes."""
    n_vars: int = 20            # Config.NVars (main.go:44)
    log_m: int = 20             # Config.LogNumConstraints (main.go:42)
    ff: int = 4                 # folding factor (mt.go:315-321: a single entry becomes [4])
    rate: int = 1               # log2(1/rho) (mt.go:322)
    security: int = 128
    pow_bits: int = 0           # taken off the query soundness (whir: protocol_security_level = security_level - pow_bits)
    soundness: str = "ConjectureList"   # | "ProvableList" | "UniqueDecoding"
    batch: int = 1              # len(proof.FirstRoundPaths) (mt.go:435)
    ood: int = 2                # OOD samples per round
    nnz_per_row: float = 2.0    # nnz(A) = nnz(B) = nnz(C) = nnz_per_row * 2^log_m
    n_statement: int = 3        # LinearStatementEvaluations: ansA, ansB, ansC (mtUtilities.go:512)
    # what a REAL params file fixes (Params.from_config: main.go:41-58); None = derived by the formulas above
    cfg_queries: tuple = None   # Config.NumQueries per round + (Config.FinalQueries,)
    cfg_ood: tuple = None       # Config.OODSamples per round
    cfg_pow: tuple = None       # Config.PowBits per round
    cfg_final_pow: int = None   # Config.FinalPowBits (mtUtilities.go:440)
    cfg_final_folding_pow: int = 0   # Config.FinalFoldingPowBits (mt.go:160)
    cfg_transcript_len: int = None   # Config.TranscriptLen (mt.go:337)

    @staticmethod
    def from_config(cfg: dict, nnz_total: int = None, batch: int = 1, **kw):
        """the census of ONE concrete configuration: ProveKit's params file as main.go:41-58 reads it (a dict of its JSON), the number of
        non-zeros of the R1CS matrices (r1cs.json: len(a.values) + len(b.values) + len(c.values), main.go:66-90) and the batch size"""
        ff = cfg["folding_factor"]
        p = Params(n_vars=cfg["n_vars"], log_m=cfg["log_num_constraints"], ff=4 if len(ff) == 1 else ff[0], rate=cfg["rate"], batch=batch,
                   cfg_queries=tuple(cfg["num_queries"]) + (cfg["final_queries"],), cfg_ood=tuple(cfg["ood_samples"]), cfg_pow=tuple(cfg["pow_bits"]),
                   cfg_final_pow=cfg.get("final_pow_bits", 0), cfg_final_folding_pow=cfg.get("final_folding_pow_bits", 0),
                   cfg_transcript_len=cfg.get("transcript_len"), n_statement=len(cfg.get("statement_evaluations", [0, 0, 0])) or 3, **kw)
        if nnz_total is not None:
            p.nnz_per_row = nnz_total / (3.0 * (1 << p.log_m))
        return p

    @property
    def n_rounds(self):         # rounds of the loop mt.go:73 (= len(RoundParametersOODSamples)); derived: every variable is folded, final_sumcheck_rounds = n_vars % ff (mt.go:317,320)
        if self.cfg_ood is not None:
            return len(self.cfg_ood)
        return (self.n_vars - self.n_vars % self.ff) // self.ff - 1

    def ood_at(self, r):
        return self.cfg_ood[r] if self.cfg_ood is not None else self.ood

    def pow_at(self, r):
        """proof-of-work bits of round r (r = n_rounds: the final PoW, mtUtilities.go:440)"""
        if self.cfg_pow is not None:
            return self.cfg_final_pow if r >= len(self.cfg_pow) else self.cfg_pow[r]
        return self.pow_bits

    @property
    def final_sumcheck_rounds(self):
        return self.n_vars % self.ff

    def queries(self, r):
        """STIR queries of round r (r = n_rounds: the final queries).  The WHIR parameter formula (whir crate, WhirConfig::queries): the code's
        rate falls by ff - 1 bits per round; ConjectureList ceil(l / log_inv_rate), ProvableList ceil(2 l / log_inv_rate), UniqueDecoding
        ceil(-l / log2((1 + rho) / 2)), l = security - pow_bits."""
        if self.cfg_queries is not None:
            return self.cfg_queries[r]
        lir = self.rate + r * (self.ff - 1)
        lam = max(0, self.security - self.pow_bits)
        if self.soundness == "ConjectureList":
            return math.ceil(lam / lir)
        if self.soundness == "ProvableList":
            return math.ceil(2 * lam / lir)
        return math.ceil(-lam / math.log2(0.5 * (1 + 2.0 ** -lir)))

    def tree_height(self, r):
        """height (levels of Compress above the leaf pair) of the tree the queries of round r open: folded domain 2^(n_vars + rate - r) / 2^ff
        (mtUtilities.go:31, mt.go:138; treeHeight = len(authPaths) + 1, mtUtilities.go:113)"""
        return self.n_vars + self.rate - r - self.ff


@dataclass
class Entry:
    term: str
    cite: str
    constraints: float = 0
    wires: dict = field(default_factory=dict)          # class -> count (new wires: internal + secret inputs)
    in_a: float = None                                 # of those wires, how many appear on some constraint's L side (pk.G1.A not infinity)
    in_b: float = None                                 # ... on some R side (pk.G1.B / pk.G2.B not infinity)
    a_cls: str = None                                  # class of the L value (the `a` vector) of these constraints
    b_cls: str = None
    public: float = 0                                  # public wires among them (not in K)
    unknown_calls: float = 0                           # third-party calls this entry only COUNTS (priced by the bucket bounds)
    bucket: str = None


# ---- third-party buckets: calls counted exactly, cost per call bounded.  lo / hi = (constraints, {class: wires}) per call.
BUCKETS = {
    # gnark-skyscraper (go.mod:10) Compress(l, r): the Skyscraper permutation on a 2-element state.  The PUBLISHED design (Bouvier et al.,
    # "Skyscraper", TCHES 2025, BN254 instance): rounds of squarings (one var x var product each) and 4 "bar" rounds that split a field
    # element into byte-sized words, push each through an 8-bit S-box and recompose.  In R1CS a bar needs the words as wires (hint),
    # their images as wires (lookup results) and a log-derivative term per looked-up row.
    #   lo: 14 squarings + 4 bars x 16 two-byte words x (word + image + one inverse)                 -> 14 + 4 x 48
    #   hi: 14 squarings + 4 bars x 32 bytes x (byte + image + row product + inverse + range-check inverse) + a bytewise canonicity check
    #       of 32 more small wires and 32 more inverses per bar                                        -> 14 + 4 x 224
    "skyscraper.Compress": {"lo": (14 + 4 * 48, {"byte": 4 * 32, "full": 14 + 4 * 16}),
                            "hi": (14 + 4 * 224, {"byte": 4 * 96, "full": 14 + 4 * 128}),
                            "basis": "published Skyscraper round structure; word size and lookup shape unknown (NewSkyscraper(api, 2), mtUtilities.go:447)"},
    # gnark-nimue (go.mod:8) Arthur over the Skyscraper sponge: a scalar read from / squeezed into the transcript costs at most one
    # permutation (rate 1) plus, for challenge BYTES, a decomposition of the squeezed element.  lo: one permutation per 2 scalars and no
    # decomposition wires; hi: one permutation per scalar (priced as Compress hi) and 32 byte wires + 32 inverses per squeezed element.
    "arthur.scalar": {"lo": (0.5 * (14 + 4 * 48), {"byte": 0.5 * 4 * 32, "full": 0.5 * (14 + 4 * 16)}),
                      "hi": (14 + 4 * 224, {"byte": 4 * 96, "full": 14 + 4 * 128}),
                      "basis": "one sponge permutation per absorbed / squeezed field element at most (NewSkyscraperArthur, mtUtilities.go:448)"},
    "arthur.challenge_bytes32": {"lo": (0.5 * (14 + 4 * 48) + 32, {"byte": 0.5 * 4 * 32 + 32, "full": 0.5 * (14 + 4 * 16)}),
                                 "hi": (14 + 4 * 224 + 64, {"byte": 4 * 96 + 32, "full": 14 + 4 * 128 + 32}),
                                 "basis": "squeeze + byte decomposition of the squeezed element (FillChallengeBytes, mtUtilities.go:35, utilities.go:82)"},
    # gnark std logderivlookup (utilities.go:189): the deferred log-derivative argument.  Per table ENTRY: a multiplicity wire (small) and
    # one inverse (full), [+ a row product for the (index, value) pair]; per QUERY: the hint's index is already counted by the caller,
    # the looked-up value is a wire, one inverse, [+ a row product].
    "logderiv.entry": {"lo": (1, {"byte": 1, "full": 1}), "hi": (3, {"byte": 1, "full": 2}), "basis": "gnark std/internal/logderivarg: multiplicity + 1 / (challenge - row)"},
    "logderiv.query": {"lo": (1, {"byte": 1, "full": 1}), "hi": (3, {"byte": 1, "full": 2}), "basis": "gnark std/lookup/logderivlookup: result wire + 1 / (challenge - row)"},
    # uints.New[U64] (mtUtilities.go:452): gnark's byte-wise XOR / AND tables (2 x 2^16 rows).  The reference only calls uapi.ToValue (a
    # linear recomposition, utilities.go:154,193; mtUtilities.go:114), so no row is ever queried: lo = nothing is emitted for an unqueried
    # table; hi = both tables are emitted whole (multiplicity + inverse per row).
    "uints.tables": {"lo": (0, {}), "hi": (2 * 65536 * 2, {"byte": 2 * 65536, "full": 2 * 65536}), "basis": "gnark std/math/uints: logderivprecomp tables of 2^16 rows"},
}


def bucket_cost(bucket, bound):
    """(constraints, {class: wires}) per call at bound 'lo' | 'hi' | 'mid' (the mean of the two)"""
    b = BUCKETS[bucket]
    if bound != "mid":
        return b[bound]
    keys = set(b["lo"][1]) | set(b["hi"][1])
    return 0.5 * (b["lo"][0] + b["hi"][0]), {k: 0.5 * (b["lo"][1].get(k, 0) + b["hi"][1].get(k, 0)) for k in keys}


def to_binary(n):
    return n + 1, {"bit": n}


def census(p: Params, hash_cost="lo"):
    """The ledger for one parameter set.  hash_cost picks the bound used for every unknown bucket ('lo' | 'mid' | 'hi')."""
    E = []
    R = p.n_rounds
    q = [p.queries(r) for r in range(R + 1)]          # q[R] = final queries
    # distinct leaves opened per round: ProveKit sorts and dedups the indices; q draws from >= 2^13 leaves collide rarely -- upper bound q
    leaves = list(q)
    n, m, ff, Bt = p.n_vars, p.log_m, p.ff, p.batch
    leaf_len = 1 << ff

    def add(term, cite, c=0, w=None, **kw):
        E.append(Entry(term, cite, c, dict(w or {}), **kw))

    def unknown(term, cite, bucket, calls):
        E.append(Entry(term, cite, unknown_calls=calls, bucket=bucket))

    # ---------------- witness inputs (no constraints): the circuit's secret / public fields
    n_scalars_transcript = (4 * m) + 2 * Bt + (R + 1) * 3 * ff + sum(1 + p.ood_at(r) for r in range(R)) + (1 << p.final_sumcheck_rounds) + 3 * p.final_sumcheck_rounds
    transcript_len = p.cfg_transcript_len if p.cfg_transcript_len is not None else 32 * n_scalars_transcript + 8 * (R + 2)
    add("Transcript bytes (PUBLIC wires)", "mtUtilities.go:92; mt.go:337-343", 0, {"byte": transcript_len}, public=transcript_len)
    add("generator, statement values / evaluations, statement points", "mtUtilities.go:65,81-84; mt.go:328-356", 0, {"full": 1 + 2 * p.n_statement, "bit": n})
    first_leaves = Bt * leaves[0]
    round_leaves = sum(leaves[1:R + 1])                # MerklePaths[r-1] is opened by round r's queries; MerklePaths[R-1] by the final queries
    add("Merkle leaves (field elements)", "mtUtilities.go:56; mt.go:287-290", 0, {"full": (first_leaves + round_leaves) * leaf_len}, in_a=0, in_b=(first_leaves + round_leaves) * leaf_len)
    path_bytes = Bt * leaves[0] * (8 + 32 * p.tree_height(0)) + sum(leaves[r] * (8 + 32 * p.tree_height(r)) for r in range(1, R + 1))
    add("leaf indexes (8 bytes), sibling hashes and auth paths (32 bytes a node)", "mtUtilities.go:57-59; mt.go:272-286", 0, {"byte": path_bytes}, in_a=0, in_b=path_bytes)

    # ---------------- initializeComponents
    unknown("uints.New tables", "mtUtilities.go:452", "uints.tables", 1)

    # ---------------- SumcheckForR1CSIOP (mtUtilities.go:354-380)
    unknown("t_rand, sp_rand, sp messages (Arthur)", "mtUtilities.go:356,367,370", "arthur.scalar", m + m * (4 + 1))
    add("R1CS-IOP sumcheck: AssertIsEqual + cubic Horner at sp_rand", "mtUtilities.go:374-376; utilities.go:24-34", 4 * m, {"full": 3 * m}, a_cls="full", b_cls="full")

    # ---------------- parseBatchedCommitment, oodAnswers, initialSumcheck
    unknown("roots, OOD point / answers, batching randomness (Arthur)", "mtUtilities.go:403-425", "arthur.scalar", 2 * Bt + 2)
    add("oodAnswers: batching powers", "mt.go:205-212", 2 * max(0, Bt - 1), {"full": 2 * max(0, Bt - 1)})
    comb0 = 1 + p.n_statement
    unknown("initial combination randomness (Arthur)", "mtUtilities.go:225", "arthur.scalar", 1)
    add("ExpandRandomness + DotProduct (initial)", "utilities.go:168-176,210-216; mtUtilities.go:156-163", 2 * (comb0 - 1), {"full": 2 * (comb0 - 1)})

    def sumcheck_rounds(k, where):
        unknown(f"sumcheck messages + folding randomness, {where} (Arthur)", "mtUtilities.go:275-281", "arthur.scalar", k * (3 + 1))
        add(f"sumcheck rounds, {where}: CheckSumOverBool + quadratic from evaluations", "mtUtilities.go:283-284; utilities.go:144-150,163-166", 4 * k, {"full": 3 * k}, a_cls="full", b_cls="full")
    sumcheck_rounds(ff, "initial")

    # ---------------- first-round leaves: batching, fold
    add("combineFirstRoundLeaves", "mtUtilities.go:467-480", max(0, Bt - 1) * (leaves[0] * leaf_len + 1), {"full": max(0, Bt - 1) * (leaves[0] * leaf_len + 1)})
    add("computeFold of the first-round answers (MultivarPoly: 2^ff - 1 products a leaf)", "mtUtilities.go:459-465; utilities.go:15-22", leaves[0] * (leaf_len - 1), {"full": leaves[0] * (leaf_len - 1)},
        a_cls="full", b_cls="full")
    add("Exponent(generator, 2^ff): squarings and running products", "mt.go:66; utilities.go:152-161", 2 * 254, {"full": 2 * 254})

    # ---------------- one Merkle multi-proof (mtUtilities.go:109-141)
    def merkle(nl, height, where):
        c, w = to_binary(height)
        add(f"Merkle {where}: ToBinary(leaf index, height)", "mtUtilities.go:114", nl * c, {k: nl * v for k, v in w.items()}, in_a=nl * height, in_b=nl * height, a_cls="bit", b_cls="bit")   # (booleanity b (1 - b) = 0 puts a bit on both sides)
        unknown(f"Merkle {where}: leaf Compress chain ({leaf_len - 1} a leaf) + one Compress a level", "mtUtilities.go:116-119,125,136", "skyscraper.Compress", nl * (leaf_len - 1 + height))
        # two Selects a level (bit x (full - full)); the And with the constant 1 is free
        add(f"Merkle {where}: left / right Select a level + root equality", "mtUtilities.go:122-123,132-134,138", nl * (2 * height + 1), {"full": nl * 2 * height}, a_cls="bit", b_cls="full")

    def is_subset(nq, nl, where):
        unknown(f"IsSubset {where}: table entries", "utilities.go:192-195", "logderiv.entry", nl)
        unknown(f"IsSubset {where}: lookups", "utilities.go:203", "logderiv.query", nq)
        add(f"IsSubset {where}: IndexOf hint outputs + AssertIsEqual", "utilities.go:199,205", nq, {"byte": nq})

    def stir(nq, where):
        nbytes = (max(1, p.tree_height(0)) + 7) // 8
        unknown(f"STIR challenge bytes {where} (Arthur)", "mtUtilities.go:34-35", "arthur.challenge_bytes32", math.ceil(nq * nbytes / 32))
        c, w = to_binary(254)
        add(f"GetStirChallenges {where}: ToBinary(full width) a query", "mtUtilities.go:48", nq * c, {k: nq * v for k, v in w.items()}, in_a=nq * 254, in_b=nq * 254, a_cls="bit", b_cls="bit")

    def exponent(nl, where):
        c, w = to_binary(254)
        add(f"Exponent a leaf {where}: ToBinary(full width)", "utilities.go:154; mt.go:101,114; mtUtilities.go:217", nl * c, {k: nl * v for k, v in w.items()}, in_a=nl * 254, in_b=nl * 254, a_cls="bit", b_cls="bit")
        add(f"Exponent a leaf {where}: product, Select, squaring a bit", "utilities.go:156-159", nl * 3 * 254, {"full": nl * 3 * 254}, a_cls="full", b_cls="full")

    def pow_check(where):
        unknown(f"PoW {where}: challenge + nonce (Arthur)", "utilities.go:82,88", "arthur.challenge_bytes32", 2)
        unknown(f"PoW {where}: Compress", "utilities.go:100", "skyscraper.Compress", 1)
        add(f"PoW {where}: AssertIsLessOrEqual against a constant", "utilities.go:132", 2 * 254 + 1, {"bit": 254, "full": 254})

    # ---------------- the round loop (mt.go:73-140)
    for r in range(R):
        unknown(f"round {r}: root, OOD points / answers (Arthur)", "mt.go:76,81; mtUtilities.go:182-186", "arthur.scalar", 1 + 2 * p.ood_at(r))
        stir(q[r], f"round {r}")
        if r == 0:
            for _ in range(Bt):
                merkle(leaves[0], p.tree_height(0), "first round")
                is_subset(q[0], leaves[0], "first round")
        else:
            merkle(leaves[r], p.tree_height(r), f"round {r}")
            is_subset(q[r], leaves[r], f"round {r}")
        exponent(leaves[r], f"round {r}")
        if p.pow_at(r) > 0:
            pow_check(f"round {r}")
        ncomb = p.ood_at(r) + leaves[r]
        unknown(f"round {r}: combination randomness (Arthur)", "mtUtilities.go:225", "arthur.scalar", 1)
        add(f"round {r}: ExpandRandomness + shift DotProduct", "mt.go:122-127; utilities.go:168-176,210-216", 2 * (ncomb - 1), {"full": 2 * (ncomb - 1)}, a_cls="full", b_cls="full")
        sumcheck_rounds(ff, f"round {r}")
        add(f"round {r}: computeFold of the next answers", "mt.go:135; utilities.go:15-22", leaves[r + 1] * (leaf_len - 1), {"full": leaves[r + 1] * (leaf_len - 1)}, a_cls="full", b_cls="full")
        add(f"round {r}: generator squared", "mt.go:139", 1, {"full": 1})

    # ---------------- final phase (mt.go:142-165; mtUtilities.go:431-444).  (The last commitment's Merkle paths are never verified in this snapshot.)
    unknown("final coefficients (Arthur)", "mtUtilities.go:433", "arthur.scalar", 1 << p.final_sumcheck_rounds)
    stir(q[R], "final")
    is_subset(q[R], leaves[R], "final")
    exponent(leaves[R], "final")
    if p.pow_at(R) > 0:
        pow_check("final")
    add("final: fold equalities", "mt.go:149-151", leaves[R], {})
    if p.final_sumcheck_rounds:
        sumcheck_rounds(p.final_sumcheck_rounds, "final")
    if p.cfg_final_folding_pow > 0:
        pow_check("final folding")

    # ---------------- ComputeWPoly (mtUtilities.go:289-326)
    def eq_outside(npts, nv, where):
        # ExpandFromUnivariate: nv squarings; EqPolyOutside: 3 products a coordinate (the first accumulation is by the constant 1); one more for the weight
        add(f"W poly: ExpandFromUnivariate + EqPolyOutside + weight, {where}", "mtUtilities.go:304,319-320; utilities.go:136-142,178-186", npts * 4 * nv, {"full": npts * 4 * nv}, a_cls="full", b_cls="full")
    eq_outside(1, n, "initial OOD query")
    nv = n
    for r in range(R):
        nv -= ff
        eq_outside(p.ood_at(r) + leaves[r], nv, f"round {r} points")
    # evaluateR1CSMatrixExtension: THE terms that make N (mtUtilities.go:494-532)
    for name, k, side in (("rows", m, "a"), ("columns", n, "b")):
        tot = (1 << (k + 1)) - 4                       # level 0 multiplies the constant 1: free; levels 1..k-1 make 2^(j+1) products each
        last = 1 << k
        # inner levels: y on the L side of both products of the next level; the final table feeds the MLE products: rows as L, columns as R
        add(f"eq table over the {name}: inner levels (y * (1 - x), y * x)", "mtUtilities.go:499-500,515-532", tot - last, {"full": tot - last}, in_a=tot - last, in_b=0, a_cls="full", b_cls="full")
        add(f"eq table over the {name}: last level (the table itself)", "mtUtilities.go:523-528", last, {"full": last}, in_a=last if side == "a" else 0, in_b=last if side == "b" else 0,
            a_cls="full", b_cls="full")
    nnz = int(3 * p.nnz_per_row * (1 << m))
    add("matrix MLE: rowEval[row] * colEval[column] a non-zero of A, B, C", "mtUtilities.go:502-510", nnz, {"full": nnz}, in_a=0, in_b=nnz, a_cls="full", b_cls="full")
    add("W poly: weights of the matrix evaluations, final equation", "mtUtilities.go:310; mt.go:177-180", p.n_statement + 2, {"full": p.n_statement + 1})
    return E, {"queries": q, "rounds": R, "transcript_len": transcript_len}


def totals(E, hash_cost):
    """-> constraints, {class: wires}, public wires, per-bucket calls, A / B membership over the entries that state it"""
    c = 0.0
    w = {k: 0.0 for k in CLASSES}
    calls = {}
    pub = 0.0
    in_a = in_b = known = 0.0
    for e in E:
        if e.bucket:
            bc, bw = bucket_cost(e.bucket, hash_cost)
            c += e.unknown_calls * bc
            for k, v in bw.items():
                w[k] += e.unknown_calls * v
            calls[e.bucket] = calls.get(e.bucket, 0) + e.unknown_calls
            continue
        c += e.constraints
        pub += e.public
        for k, v in e.wires.items():
            w[k] += v
        if e.in_a is not None:
            in_a += e.in_a; in_b += e.in_b; known += sum(e.wires.values())
    return c, w, pub, calls, (in_a, in_b, known)


def mix_of(w):
    t = sum(w.values())
    return {k: w[k] / t for k in CLASSES}


def scenarios():
    """The scenario grid for BASELINE configs[1]: every combination of the axes BASELINE.json leaves open."""
    out = []
    for log_m in (17, 18, 19, 20):
        for nnz_row in (1.0, 2.0, 3.0):
            for snd, powb in (("ConjectureList", 0), ("ConjectureList", 20), ("ProvableList", 0), ("UniqueDecoding", 0)):
                for batch in (1, 2):
                    for hc in ("lo", "mid", "hi"):
                        p = Params(log_m=log_m, nnz_per_row=nnz_row, soundness=snd, pow_bits=powb, batch=batch)
                        E, info = census(p, hc)
                        c, w, pub, calls, ab = totals(E, hc)
                        out.append({"params": p, "hash_cost": hc, "constraints": c, "wires": w, "mix": mix_of(w), "calls": calls, "info": info, "ab": ab})
    return out


def census_range(target_log_n=23):
    """Scenarios whose constraint count pads to the FFT domain 2^target_log_n (configs[1] -> N = 2^23) and the range of their
    mixes.  -> (feasible scenarios, {class: (min, max)}, midpoint mix in per-mille for the generators: bit, byte, u64; the rest uniform)"""
    S = [s for s in scenarios() if (1 << (target_log_n - 1)) < s["constraints"] <= (1 << target_log_n)]
    rng = {k: (min(s["mix"][k] for s in S), max(s["mix"][k] for s in S)) for k in CLASSES}
    mid = {k: 0.5 * (rng[k][0] + rng[k][1]) for k in CLASSES}
    t = sum(mid.values())
    mid = {k: v / t for k, v in mid.items()}
    pm = {k: int(round(1000 * mid[k])) for k in ("bit", "byte", "u64")}
    return S, rng, mid, pm


def census_mix_permille():
    """(bit, byte, u64) per-mille of the census's midpoint mix; the rest is full-width.  What `bench.py --dist census` feeds MI_DIST_MIX."""
    return tuple(census_range()[3][k] for k in ("bit", "byte", "u64"))


def report():
    L = []
    P = L.append
    p0 = Params()
    P("WIRE-CLASS CENSUS of the WHIR-verifier circuit (tools/wire_census.py; counting rules restated from /root/reference, cited per term)")
    P("=" * 150)
    P(f"reference point: n_vars {p0.n_vars}, folding factor {p0.ff}, rate 1/2, {p0.security}-bit {p0.soundness}, log_num_constraints {p0.log_m}, {p0.nnz_per_row} non-zeros a row and matrix, batch {p0.batch}")
    for hc in ("lo", "hi"):
        E, info = census(p0, hc)
        c, w, pub, calls, ab = totals(E, hc)
        if hc == "lo":
            P(f"rounds {info['rounds']}, queries per round {info['queries'][:-1]} + final {info['queries'][-1]}, transcript {info['transcript_len']} bytes (public wires)")
            P("")
            P(f"{'term':<92} {'cite':<44} {'constraints':>12} {'bit':>10} {'byte':>10} {'u64':>6} {'full':>11}")
            for e in E:
                if e.bucket:
                    P(f"{e.term:<92.92} {e.cite:<44.44} {'UNKNOWN: ' + e.bucket + ' x ' + format(e.unknown_calls, 'g'):>52}")
                else:
                    P(f"{e.term:<92.92} {e.cite:<44.44} {e.constraints:>12,.0f} {e.wires.get('bit', 0):>10,.0f} {e.wires.get('byte', 0):>10,.0f} {e.wires.get('u64', 0):>6,.0f} {e.wires.get('full', 0):>11,.0f}")
            P("")
            P("unknown buckets (third-party code absent from the container; CALLS are exact, the cost per call is bounded, not recalled):")
            for b, k in calls.items():
                lo, hi = BUCKETS[b]["lo"], BUCKETS[b]["hi"]
                P(f"  {b:<28} calls {k:>10,.0f}   per call: lo {lo[0]:g} constraints {lo[1]}   hi {hi[0]:g} constraints {hi[1]}")
                P(f"  {'':<28} basis: {BUCKETS[b]['basis']}")
            P("")
        m_ = mix_of(w)
        P(f"[{hc}] constraints {c:,.0f} = 2^{math.log2(c):.2f}; wires by class: " + ", ".join(f"{k} {w[k]:,.0f} ({100 * m_[k]:.1f} %)" for k in CLASSES))
        if hc == "lo":
            known_terms = [e for e in E if not e.bucket and e.in_a is not None]
            P(f"      L / R membership over the terms that state it ({ab[2]:,.0f} wires): on an L side (pk.G1.A not infinity) {100 * ab[0] / ab[2]:.0f} %, on an R side (pk.G1.B / G2.B not infinity) {100 * ab[1] / ab[2]:.0f} %"
              f"  -- the benchmark's masks say 90 % / 50 % (BASELINE.md 3)")
    P("")
    P("WHICH TERMS MAKE N: the eq tables (2^(n_vars+1) + 2^(log_m+1) full-width products, mtUtilities.go:515-532) and the matrix MLE (one full-width product a non-zero,")
    P("mtUtilities.go:502-510) are exact and ALL full-width; the Merkle / STIR / PoW part is thousands of Compress calls whose inside is bytes and full-width values in")
    P("comparable numbers; bits (ToBinary at mtUtilities.go:48,114, utilities.go:154) are 1-3 % of the wires.  No term of the circuit produces 64-bit values:")
    P("uints.U64 is eight byte wires (utilities.go:154,193 recompose them linearly).")
    P("")
    S, rng, mid, pm = census_range()
    P(f"SCENARIO GRID for BASELINE configs[1] (log_m 17..20 x non-zeros a row 1..3 x soundness / PoW x batch 1..2 x bucket bound lo / mid / hi = {len(scenarios())} scenarios);")
    P(f"kept: the {len(S)} whose constraint count pads to N = 2^23 (BASELINE.md 3):")
    P(f"{'log_m':>5} {'nnz/row':>7} {'soundness':<15} {'pow':>3} {'batch':>5} {'bound':>5} {'constraints':>12} {'bit %':>6} {'byte %':>6} {'u64 %':>6} {'full %':>6} {'Compress calls':>14}")
    for s in S:
        p = s["params"]
        P(f"{p.log_m:>5} {p.nnz_per_row:>7.1f} {p.soundness:<15} {p.pow_bits:>3} {p.batch:>5} {s['hash_cost']:>5} {s['constraints']:>12,.0f} " +
          " ".join(f"{100 * s['mix'][k]:>6.1f}" for k in CLASSES) + f" {s['calls'].get('skyscraper.Compress', 0):>14,.0f}")
    P("")
    P("RANGE of the mix over the kept scenarios: " + ", ".join(f"{k} {100 * rng[k][0]:.1f}-{100 * rng[k][1]:.1f} %" for k in CLASSES))
    P("MIDPOINT (normalised): " + ", ".join(f"{k} {100 * mid[k]:.1f} %" for k in CLASSES) + f"   ->  MI_DIST_MIX per-mille (bit, byte, u64) = ({pm['bit']}, {pm['byte']}, {pm['u64']}), the rest full-width")
    P("BASELINE.md 3's guess, for comparison: bit 45 %, byte 25 %, u64 5 %, full 25 %  -- the full-width share of every kept scenario is above it.")
    return "\n".join(L)


def report_config(cfg, nnz_total, batch):
    """the census of ONE configuration (a params file): the ledger's totals under the three bucket bounds, and the MI_DIST_MIX to bench with"""
    L = []
    p = Params.from_config(cfg, nnz_total=nnz_total, batch=batch)
    L.append(f"params: n_vars {p.n_vars}, log_num_constraints {p.log_m}, folding factor {p.ff}, rate {p.rate}, rounds {p.n_rounds}, queries {list(p.cfg_queries)}, "
             f"OOD samples {list(p.cfg_ood)}, PoW bits {list(p.cfg_pow)} + {p.cfg_final_pow}, batch {batch}, non-zeros of A + B + C {int(3 * p.nnz_per_row * (1 << p.log_m)):,}"
             + ("" if nnz_total is not None else "  (ASSUMED: 2 a row and matrix -- pass --r1cs for the real count)"))
    mixes = []
    for hc in ("lo", "mid", "hi"):
        E, _ = census(p, hc)
        c, w, pub, calls, ab = totals(E, hc)
        m_ = mix_of(w)
        mixes.append(m_)
        L.append(f"[{hc:>3}] constraints {c:>12,.0f} = 2^{math.log2(c):.2f} -> FFT domain 2^{math.ceil(math.log2(c))};  " + ", ".join(f"{k} {100 * m_[k]:.1f} %" for k in CLASSES) +
                 f";  Compress calls {calls.get('skyscraper.Compress', 0):,.0f}")
    mid = mixes[1]
    L.append(f"bench it: python bench.py --log-n <the domain above> --dist mix:{round(1000 * mid['bit'])},{round(1000 * mid['byte'])},{round(1000 * mid['u64'])}   (MI_DIST_MIX per-mille of bits, bytes, 64-bit; the rest full-width)")
    return "\n".join(L)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--write", default="")
    ap.add_argument("--json", action="store_true")
    ap.add_argument("--params", default="", help="ProveKit's params file (the JSON main.go:41-58 reads): the census of THAT configuration instead of the scenario grid")
    ap.add_argument("--r1cs", default="", help="with --params: ProveKit's r1cs.json (main.go:82-90): the real number of non-zeros of A, B, C")
    ap.add_argument("--batch", type=int, default=1, help="with --params: len(round0_merkle_paths) of the proof (mt.go:435)")
    a = ap.parse_args()
    if a.params:
        cfg = json.load(open(a.params))
        nnz = None
        if a.r1cs:
            r = json.load(open(a.r1cs))
            nnz = sum(len(r[k]["values"]) for k in ("a", "b", "c"))
        print(report_config(cfg, nnz, a.batch))
        sys.exit(0)
    if a.json:
        S, rng, mid, pm = census_range()
        print(json.dumps({"range": rng, "midpoint": mid, "permille": pm, "scenarios_kept": len(S)}))
        sys.exit(0)
    txt = report()
    if a.write:
        with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), a.write), "w") as f:
            f.write(txt + "\n")
    print(txt)
