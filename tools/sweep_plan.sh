#!/bin/bash
# sweep of the item lengths (L1, L2) of the level machinery on one box: bench.py --msm-plan 0,L1,L2,0,0
for plan in "$@"; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --sharded-msm-log-n 0 --steps 20 --msm-plan $plan > gpurun_out/sweep_$plan.log 2>&1 || exit 1
  python - <<PY
import json
l = json.loads(open("gpurun_out/sweep_$plan.log").read().strip().splitlines()[-1])
print("plan $plan", "proofs/s %.2f" % l["value"], "latency %.2f" % l["single_proof_latency_ms"], "accum launch %.3f" % l["roofline"]["launch_ms"], flush=True)
PY
done
