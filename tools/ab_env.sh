#!/bin/bash
# A/B of environment settings on ONE box: tools/ab_env.sh ROUNDS "VAR=val [VAR2=val2]" ...   (the empty setting "" is the baseline)
rounds=$1; shift
for r in $(seq 1 $rounds); do
  for setting in "" "$@"; do
    env $setting timeout -k 10 200 python bench.py --no-cpu-baseline --no-host-inputs --sharded-msm-log-n 0 --sharded-prove-log-n 0 --steps 20 > gpurun_out/abe.log 2>&1 || exit 1
    python - <<PY
import json
l = json.loads(open("gpurun_out/abe.log").read().strip().splitlines()[-1])
print("r$r [$setting]", "proofs/s %.2f" % l["value"], "latency %.2f" % l["single_proof_latency_ms"], "phases", {k: round(v, 1) for k, v in l["phase_ms"].items() if k != "assemble_ms"}, flush=True)
PY
  done
done
