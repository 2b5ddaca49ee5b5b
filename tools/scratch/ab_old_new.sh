#!/bin/bash
# same-box comparison of the round-2 tree (old_r2/, a git worktree of f46faf9) with the current one: bench.py --steps 20, host-input leg included
for r in 1 2; do
  (cd old_r2 && timeout -k 10 200 python bench.py --no-cpu-baseline --sharded-msm-log-n 0 --steps 20 > ../gpurun_out/abo.log 2>&1) || exit 1
  python - <<PY
import json
l = json.loads(open("gpurun_out/abo.log").read().strip().splitlines()[-1])
print("r$r OLD", "proofs/s %.2f" % l["value"], "host inputs %.2f" % l["value_host_inputs"], "ratio %.4f" % (l["value_host_inputs"] / l["value"]), "latency %.2f" % l["single_proof_latency_ms"], flush=True)
PY
  timeout -k 10 200 python bench.py --no-cpu-baseline --sharded-msm-log-n 0 --sharded-prove-log-n 0 --steps 20 > gpurun_out/abn.log 2>&1 || exit 1
  python - <<PY
import json
l = json.loads(open("gpurun_out/abn.log").read().strip().splitlines()[-1])
print("r$r NEW", "proofs/s %.2f" % l["value"], "host inputs %.2f" % l["value_host_inputs"], "ratio %.4f" % (l["value_host_inputs"] / l["value"]), "latency %.2f" % l["single_proof_latency_ms"], flush=True)
PY
done
