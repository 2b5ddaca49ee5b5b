#!/bin/bash
# A/B of environment settings on ONE box: tools/ab_env.sh ROUNDS "VAR=val [VAR2=val2]" ...   (the empty setting "" is the baseline)
rounds=$1; shift
for r in $(seq 1 $rounds); do
  for setting in "" "$@"; do
    env $setting timeout -k 10 300 python bench.py --no-cpu-baseline --sharded-msm-log-n 0 --sharded-prove-log-n 0 --steps 20 > gpurun_out/abe.log 2>&1 || exit 1
    python - <<PY
import json
l = json.loads([x for x in open("gpurun_out/abe.log") if x.startswith("{")][-1])
print("r$r [$setting]", "value %.2f" % l["value"], "hbm %.2f" % l["value_hbm_resident_inputs"], "lat %.2f" % l["single_proof_latency_ms"], "lat_host %.2f" % l["single_proof_latency_host_inputs_ms"],
      "accum launch %.2f ms frac %.4f" % (l["roofline"]["launch_ms"], l["roofline"]["frac"]), flush=True)
PY
  done
done
