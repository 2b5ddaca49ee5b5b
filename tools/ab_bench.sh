#!/bin/bash
# A/B of two builds on ONE box: the in-tree library against gnark-whir_amd/build_ab/libab.so (bench.py through MI355X_GROTH16_LIB).
# usage (on the GPU box): tools/ab_bench.sh [rounds] [extra bench.py args...]
rounds=${1:-2}; shift
for r in $(seq 1 $rounds); do
  for v in A B; do
    if [ $v = B ]; then export MI355X_GROTH16_LIB=$PWD/gnark-whir_amd/build_ab/libab.so; else unset MI355X_GROTH16_LIB; fi
    timeout -k 10 200 python bench.py --no-cpu-baseline --sharded-msm-log-n 0 --sharded-prove-log-n 0 --no-sensitivity --steps 24 "$@" > gpurun_out/ab_$v$r.log 2>&1 || exit 1
    python - <<PY
import json
l = json.loads(open("gpurun_out/ab_$v$r.log").read().strip().splitlines()[-1])
print("$v$r", "proofs/s (host inputs, 196 B) %.2f" % l["value"], "HBM-resident %.2f" % l["value_hbm_resident_inputs"], "latency %.2f" % l["single_proof_latency_ms"], "computeH solo %.3f" % l["roofline_ntt"]["compute_h_solo_ms"],
      "accum launch %.3f" % l["roofline"]["launch_ms"], "solo adds/s %.3e" % l["g1_msm_solo"]["mixed_adds_per_s"], "modmul/s %.3e" % l["valu"]["modmul_ceiling_per_s"], flush=True)
PY
  done
done
