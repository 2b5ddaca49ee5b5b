#!/bin/bash
# A/B on ONE box of the host-input leg (bench.py --steps 20): helper threads x early handover.   usage: tools/ab_host_path.sh ROUNDS
for r in $(seq 1 $1); do
  for v in "1 1" "0 1" "0 0" "1 0"; do
    set -- $v
    MI_PROVE_HELPER_THREADS=$1 MI_POOL_EARLY_HANDOVER=$2 timeout -k 10 200 python bench.py --no-cpu-baseline --sharded-msm-log-n 0 --sharded-prove-log-n 0 --steps 20 > gpurun_out/abh.log 2>&1 || exit 1
    python - <<PY
import json
l = json.loads(open("gpurun_out/abh.log").read().strip().splitlines()[-1])
print("r$r threads=$1 early=$2", "proofs/s %.2f" % l["value"], "HBM-resident %.2f" % l["value_hbm_resident_inputs"], "ratio %.4f" % (l["value"] / l["value_hbm_resident_inputs"]), "latency %.2f" % l["single_proof_latency_ms"], flush=True)
PY
  done
done
