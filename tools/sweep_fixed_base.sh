#!/bin/bash
# sweep of the fixed-base window widths on one box: bench.py --fixed-base c_ak,c_b,c_z
for fb in "$@"; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --sharded-msm-log-n 0 --sharded-prove-log-n 0 --steps 20 --fixed-base $fb > gpurun_out/sweepfb_$fb.log 2>&1 || exit 1
  python - <<PY
import json
l = json.loads(open("gpurun_out/sweepfb_$fb.log").read().strip().splitlines()[-1])
print("fixed-base $fb", "proofs/s %.2f" % l["value"], "latency %.2f" % l["single_proof_latency_ms"], "hbm %.1f GB" % l["hbm_in_use_gb"], "pk load %.1f s" % l["pk_load_s"], flush=True)
PY
done
