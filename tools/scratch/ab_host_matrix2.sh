#!/bin/bash
for r in 1 2; do
  for v in "0 0" "1 0" "1 1"; do
    set -- $v
    MI_PROVE_OLD_ORDER=$1 MI_POOL_EARLY_HANDOVER=$2 timeout -k 10 200 python bench.py --no-cpu-baseline --sharded-msm-log-n 0 --sharded-prove-log-n 0 --steps 20 > gpurun_out/abh.log 2>&1 || exit 1
    python - <<PY
import json
l = json.loads(open("gpurun_out/abh.log").read().strip().splitlines()[-1])
print("r$r old_order=$1 early=$2", "proofs/s %.2f" % l["value"], "host inputs %.2f" % l["value_host_inputs"], "ratio %.4f" % (l["value_host_inputs"] / l["value"]), "latency %.2f" % l["single_proof_latency_ms"], "upload", l["host_inputs_upload_ms"], flush=True)
PY
  done
done
