#!/bin/bash
# Why does the FULL default bench line show a lower `value` / HBM-resident ratio than the reduced A/B runs?  Same box, one process per run.
for r in 1 2; do
  i=0
  for v in "" "--no-cpu-baseline" "--sharded-msm-log-n 0 --sharded-prove-log-n 0" "--no-cpu-baseline --sharded-msm-log-n 0 --sharded-prove-log-n 0"; do
    timeout -k 10 400 python bench.py --no-sensitivity --no-live-pmc $v > gpurun_out/dvr.log 2>&1 || { tail -3 gpurun_out/dvr.log; exit 1; }
    python3 - <<PY
import json
l = json.loads([x for x in open("gpurun_out/dvr.log") if x.startswith("{")][-1])
print("r$r v$i [$v]", "value %.2f" % l["value"], "hbm %.2f" % l["value_hbm_resident_inputs"], "ratio %.3f" % (l["value"] / l["value_hbm_resident_inputs"]), "h2d", l["host_inputs_upload_ms"], "lat_host %.1f" % l["single_proof_latency_host_inputs_ms"], flush=True)
PY
    i=$((i+1))
  done
done
