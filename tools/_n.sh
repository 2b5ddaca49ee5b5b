set -e
for pl in 9,9,7,256 9,9,7,512 9,9,7,128 10,10,7,256 10,10,7,512 9,9,8,256 8,8,8,256 10,10,8,512 11,11,6,512 10,9,7,256; do
python bench.py --no-cpu-baseline --ntt-plan $pl > gpurun_out/n_$pl.log 2>&1 || { echo "$pl failed"; tail -2 gpurun_out/n_$pl.log; continue; }
python - <<PY
import json
for l in open("gpurun_out/n_$pl.log"):
    if l.startswith("{"):
        d=json.loads(l); print("$pl", round(d["value"],2), round(d["ms_per_step"],2), round(d["single_proof_latency_ms"],2), flush=True)
PY
done
