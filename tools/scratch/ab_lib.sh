#!/bin/bash
# current bench.py against two builds of the library on ONE box (host-input leg included).   usage: tools/scratch/ab_lib.sh LIB_B [rounds]
for r in $(seq 1 ${2:-2}); do
  for v in A B; do
    if [ $v = B ]; then export MI355X_GROTH16_LIB=$PWD/$1; else unset MI355X_GROTH16_LIB; fi
    timeout -k 10 200 python bench.py --no-cpu-baseline --sharded-msm-log-n 0 --sharded-prove-log-n 0 --steps 20 > gpurun_out/abl.log 2>&1 || { tail -3 gpurun_out/abl.log; exit 1; }
    python - <<PY
import json
l = json.loads(open("gpurun_out/abl.log").read().strip().splitlines()[-1])
print("r$r $v", "proofs/s %.2f" % l["value"], "HBM-resident %.2f" % l["value_hbm_resident_inputs"], "ratio %.4f" % (l["value"] / l["value_hbm_resident_inputs"]), "latency %.2f" % l["single_proof_latency_ms"], "upload", l.get("host_inputs_upload_ms"), flush=True)
PY
  done
done
