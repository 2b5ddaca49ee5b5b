#!/bin/bash
# Same-box A/B of bench.py argument sets, one process per run, alternating: tools/ab_args.sh ROUNDS "args of variant 0" "args of variant 1" ...
rounds=$1; shift
for r in $(seq 1 $rounds); do
  i=0
  for v in "$@"; do
    timeout -k 10 200 python bench.py --no-cpu-baseline --sharded-msm-log-n 0 --sharded-prove-log-n 0 --no-sensitivity --no-live-pmc --steps 30 $v > gpurun_out/aba.log 2>&1 || { tail -3 gpurun_out/aba.log; exit 1; }
    python3 - <<PY
import json
l = json.loads([x for x in open("gpurun_out/aba.log") if x.startswith("{")][-1])
print("r$r v$i [$v]", "value %.2f" % l["value"], "hbm %.2f" % l["value_hbm_resident_inputs"], "lat %.2f" % l["single_proof_latency_ms"], "computeH solo %.3f" % l["roofline_ntt"]["compute_h_solo_ms"], flush=True)
PY
    i=$((i+1))
  done
done
