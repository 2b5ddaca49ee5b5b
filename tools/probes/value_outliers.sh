#!/bin/bash
# Repeats the reduced bench run and prints value, the HBM-resident rate and the step-completion gaps: what does a depressed `value` look like?
for r in $(seq 1 ${1:-10}); do
  timeout -k 10 200 python bench.py --no-cpu-baseline --sharded-msm-log-n 0 --sharded-prove-log-n 0 --no-sensitivity --no-live-pmc --no-solo-legs > gpurun_out/vo.log 2>&1 || { tail -3 gpurun_out/vo.log; exit 1; }
  python3 - <<PY
import json
l = json.loads([x for x in open("gpurun_out/vo.log") if x.startswith("{")][-1])
g = l["step_completion_gaps"]
print("r$r value %.2f hbm %.2f ratio %.3f gaps median %.1f p90 %.1f max %.1f" % (l["value"], l["value_hbm_resident_inputs"], l["value"] / l["value_hbm_resident_inputs"], g["median_ms"], g["p90_ms"], g["max_ms"]), "h2d max %.1f" % l["host_inputs_upload_ms"]["max"], flush=True)
PY
done
