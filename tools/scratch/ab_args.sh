#!/bin/bash
# A/B of bench.py argument sets on ONE box: tools/scratch/ab_args.sh ROUNDS "ARGS A" "ARGS B" ...
rounds=$1; shift
for r in $(seq 1 $rounds); do
  for a in "$@"; do
    timeout -k 10 300 python bench.py --no-cpu-baseline --sharded-msm-log-n 0 --sharded-prove-log-n 0 --steps 24 $a > gpurun_out/aba.log 2>&1 || { tail -3 gpurun_out/aba.log; exit 1; }
    python - <<PY
import json
l = json.loads([x for x in open("gpurun_out/aba.log") if x.startswith("{")][-1])
print("r$r [$a]", "value %.2f" % l["value"], "hbm %.2f" % l["value_hbm_resident_inputs"], "lat %.2f" % l["single_proof_latency_ms"], "lat_host %.2f" % l["single_proof_latency_host_inputs_ms"], "accum launch %.2f ms" % l["roofline"]["launch_ms"], flush=True)
PY
  done
done
