set -e
timeout -k 10 400 python -m pytest tests -x -q -m gpu > gpurun_out/t_tests.log 2>&1; tail -2 gpurun_out/t_tests.log
python bench.py --no-cpu-baseline --steps 30 > gpurun_out/t_def.log 2>&1
python - <<PY
import json
for l in open("gpurun_out/t_def.log"):
    if l.startswith("{"):
        d=json.loads(l); print("default", round(d["value"],2), round(d["ms_per_step"],2), round(d["single_proof_latency_ms"],2), flush=True)
PY
