#!/bin/bash
# (chunk, group bits) of the fixed-base sort on ONE box: tools/scratch/sweep_sort.sh "CHUNK:GBITS" ...   (0 = the default)
for cfg in "$@"; do
  ch=${cfg%%:*}; gb=${cfg##*:}
  timeout -k 10 300 python bench.py --no-cpu-baseline --sharded-msm-log-n 0 --sharded-prove-log-n 0 --steps 24 --msm-chunk $ch --msm-group-bits $gb > gpurun_out/sws.log 2>&1 || { tail -3 gpurun_out/sws.log; exit 1; }
  python - <<PY
import json
l = json.loads([x for x in open("gpurun_out/sws.log") if x.startswith("{")][-1])
print("chunk $ch gbits $gb", "value %.2f" % l["value"], "hbm %.2f" % l["value_hbm_resident_inputs"], "lat %.2f" % l["single_proof_latency_ms"], flush=True)
PY
done
