"""CPU suite: tools/wire_census.py -- the wire-class census of the reference's circuit that stands behind `bench.py --dist census` /
`value_census_mix`.  The tool restates counting rules; these tests pin the arithmetic of the big exact terms against
their closed forms (mtUtilities.go:494-532) and the properties the README quotes."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import wire_census as wc  # noqa: E402


def _emulated_eq_table(k):
    """calculateEQOverBooleanHypercube (mtUtilities.go:515-532) under gnark's rule 'a product with a constant operand is free':
    count the var x var products by running the loop on symbols"""
    ans = ["1"]
    n = 0
    for _ in range(k):
        nxt = []
        for y in ans:
            n += 0 if y == "1" else 2          # y * (1 - x) and y * x
            nxt += ["v", "v"]
        ans = nxt
    return n, len(ans)


def test_eq_table_terms_equal_the_emulated_loop():
    for k in (1, 2, 5, 12):
        prods, size = _emulated_eq_table(k)
        assert size == 1 << k and prods == (1 << (k + 1)) - 4 if k > 1 else prods == 0
    p = wc.Params(log_m=12, n_vars=12)
    E, _ = wc.census(p)
    rows = sum(e.constraints for e in E if e.term.startswith("eq table over the rows"))
    cols = sum(e.constraints for e in E if e.term.startswith("eq table over the columns"))
    assert rows == cols == _emulated_eq_table(12)[0]
    mle = [e for e in E if e.term.startswith("matrix MLE")][0]
    assert mle.constraints == int(3 * p.nnz_per_row * (1 << 12)) and mle.wires == {"full": mle.constraints}


def test_queries_follow_the_whir_formula():
    p = wc.Params()
    assert [p.queries(r) for r in range(5)] == [128, 32, 19, 13, 10]            # ceil(128 / (1 + 3 r))
    assert wc.Params(soundness="ProvableList").queries(0) == 256
    assert wc.Params(pow_bits=20).queries(0) == 108
    assert p.n_rounds == 4 and p.final_sumcheck_rounds == 0 and [p.tree_height(r) for r in range(4)] == [17, 16, 15, 14]


def test_every_kept_scenario_is_mostly_full_width_and_the_midpoint_is_what_bench_uses():
    S, rng, mid, pm = wc.census_range()
    assert len(S) >= 20
    for s in S:
        assert (1 << 22) < s["constraints"] <= (1 << 23)
        assert s["mix"]["full"] > 0.5 > 0.25, "BASELINE.md 3's guess (25 % full-width) is below every scenario of the census"
        assert s["mix"]["u64"] == 0.0 and s["mix"]["bit"] < 0.05
    assert wc.census_mix_permille() == (pm["bit"], pm["byte"], pm["u64"]) and sum(wc.census_mix_permille()) < 500
    assert 0.6 < mid["full"] < 0.9


def test_every_term_cites_the_reference_and_unknowns_are_bounded_not_priced():
    E, _ = wc.census(wc.Params())
    for e in E:
        assert ".go:" in e.cite, e.term
        if e.bucket:
            b = wc.BUCKETS[e.bucket]
            assert b["lo"][0] <= b["hi"][0] and b["basis"] and e.constraints == 0 and not e.wires
    txt = wc.report()
    committed = open(os.path.join(ROOT, "profiles", "r06_wire_census.txt")).read()
    assert txt.strip() == committed.strip(), "profiles/r06_wire_census.txt is not what tools/wire_census.py prints: python tools/wire_census.py --write profiles/r06_wire_census.txt"


def test_census_of_a_concrete_params_file():
    """`tools/wire_census.py --params FILE`: the census of ONE configuration as ProveKit's params file states it (main.go:41-58) -- queries,
    OOD samples and PoW bits per round taken from the file, not from the WHIR formulas; here the committed params fixture of the ingestion tests"""
    import json
    cfg = json.loads(json.load(open(os.path.join(ROOT, "tests", "golden", "whir_params_small.json")))["text"])
    p = wc.Params.from_config(cfg, nnz_total=3 * 5 * (1 << 17), batch=2)
    assert p.n_rounds == 4 and [p.queries(r) for r in range(5)] == [103, 46, 30, 23, 18] and [p.ood_at(r) for r in range(4)] == [2, 2, 1, 1]
    assert [p.pow_at(r) for r in range(5)] == [18, 20, 22, 22, 21] and abs(p.nnz_per_row - 5.0) < 1e-9
    E, info = wc.census(p, "lo")
    calls = wc.totals(E, "lo")[3]
    # Merkle: 2 first-round trees x 103 leaves x (15 + 17) + 46 x (15 + 16) + 30 x (15 + 15) + 23 x (15 + 14); PoW: one Compress a round + final
    assert calls["skyscraper.Compress"] == 2 * 103 * 32 + 46 * 31 + 30 * 30 + 23 * 29 + 5
    assert info["transcript_len"] == cfg["transcript_len"]
    txt = wc.report_config(cfg, None, 1)
    assert "ASSUMED" in txt and "--dist mix:" in txt
