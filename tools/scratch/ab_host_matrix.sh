#!/bin/bash
# host-input leg on ONE box: helper threads x early hand-over (the not-early jobs run the plain device-pointer path)
for r in 1 2; do
  for v in "1 1" "0 0" "1 0" "0 1"; do
    set -- $v
    MI_PROVE_HELPER_THREADS=$1 MI_POOL_EARLY_HANDOVER=$2 timeout -k 10 200 python bench.py --no-cpu-baseline --sharded-msm-log-n 0 --sharded-prove-log-n 0 --steps 20 > gpurun_out/abh.log 2>&1 || exit 1
    python - <<PY
import json
l = json.loads(open("gpurun_out/abh.log").read().strip().splitlines()[-1])
print("r$r threads=$1 early=$2", "proofs/s %.2f" % l["value"], "host inputs %.2f" % l["value_host_inputs"], "ratio %.4f" % (l["value_host_inputs"] / l["value"]), "upload", l["host_inputs_upload_ms"], flush=True)
PY
  done
done
