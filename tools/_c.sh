set -e
for fb in 19,18,20 20,18,20 20,19,21 19,18,21 20,19,20 21,19,21 18,17,19 20,20,20; do
python bench.py --no-cpu-baseline --steps 30 --fixed-base $fb > gpurun_out/c_$fb.log 2>&1
python - <<PY
import json
for l in open("gpurun_out/c_$fb.log"):
    if l.startswith("{"):
        d=json.loads(l); print("$fb", round(d["value"],2), round(d["ms_per_step"],2), round(d["single_proof_latency_ms"],2), d["hbm_in_use_gb"], flush=True)
PY
done
