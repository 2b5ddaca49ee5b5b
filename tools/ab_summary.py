"""Summary of tools/ab_bench.sh's logs (gpurun_out/ab_A<r>.log / ab_B<r>.log): means, difference, rounds won.  python tools/ab_summary.py ROUNDS"""
import json
import statistics as st
import sys
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
A, B = [], []
for r in range(1, n + 1):
    for v, L in (("A", A), ("B", B)):
        l = json.loads(open(f"gpurun_out/ab_{v}{r}.log").read().strip().splitlines()[-1])
        L.append((l["value"], l["value_hbm_resident_inputs"], l["single_proof_latency_ms"]))
for i, name in enumerate(("value", "hbm-resident", "single proof ms")):
    a = [x[i] for x in A]; b = [x[i] for x in B]
    wins = sum(1 for x, y in zip(a, b) if (x > y if i < 2 else x < y))
    print(f"{name}: A {st.mean(a):.2f}  B {st.mean(b):.2f}  A/B {100 * (st.mean(a) / st.mean(b) - 1):+.2f} %  A wins {wins}/{len(a)}")
