#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for r in 1 2; do for c in 4 5 6 8; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-hbm-resident --sharded-msm-log-n 0 --sharded-prove-log-n 0 --steps 30 --callers $c > gpurun_out/abc.log 2>&1 || exit 1
  python - <<PY
import json
l = json.loads([x for x in open("gpurun_out/abc.log") if x.startswith("{")][-1])
print("r$r callers $c value %.2f ms/step %.2f upload median %.1f" % (l["value"], l["ms_per_step"], l["host_inputs_upload_ms"]["median"]), flush=True)
PY
done; done
