#!/bin/bash
# in-flight sweep on ONE box: tools/scratch/ab_inflight.sh ROUNDS N1 N2 ...
rounds=$1; shift
for r in $(seq 1 $rounds); do
  for k in "$@"; do
    timeout -k 10 300 python bench.py --no-cpu-baseline --sharded-msm-log-n 0 --sharded-prove-log-n 0 --steps 24 --in-flight $k > gpurun_out/abi.log 2>&1 || exit 1
    python - <<PY
import json
l = json.loads([x for x in open("gpurun_out/abi.log") if x.startswith("{")][-1])
print("r$r in-flight $k", "value %.2f" % l["value"], "hbm %.2f" % l["value_hbm_resident_inputs"], "accum launch %.2f ms" % l["roofline"]["launch_ms"], "hbm GB %.1f" % l["hbm_in_use_gb"], flush=True)
PY
  done
done
