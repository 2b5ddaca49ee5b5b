#!/bin/bash
# A/B on ONE box of the pool's early handover (a host job goes to a worker once W has arrived) against the full-upload handover:
# the host-input leg of bench.py at the driver's --steps 20.   usage: tools/ab_pool_handover.sh ROUNDS
for r in $(seq 1 $1); do
  for v in 1 0; do
    MI_POOL_EARLY_HANDOVER=$v timeout -k 10 200 python bench.py --no-cpu-baseline --sharded-msm-log-n 0 --sharded-prove-log-n 0 --steps 20 > gpurun_out/abh.log 2>&1 || exit 1
    python - <<PY
import json
l = json.loads(open("gpurun_out/abh.log").read().strip().splitlines()[-1])
print("r$r early_handover=$v", "proofs/s %.2f" % l["value"], "HBM-resident %.2f" % l["value_hbm_resident_inputs"], "ratio %.4f" % (l["value"] / l["value_hbm_resident_inputs"]), "latency %.2f" % l["single_proof_latency_ms"], flush=True)
PY
  done
done
