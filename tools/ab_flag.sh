#!/bin/bash
# A/B of one bench.py flag on ONE box: alternating runs without / with it.   usage: tools/ab_flag.sh ROUNDS FLAG [more bench.py args]
rounds=$1; flag=$2; shift 2
for r in $(seq 1 $rounds); do
  for v in A B; do
    extra=""; [ $v = B ] && extra=$flag
    timeout -k 10 200 python bench.py --no-cpu-baseline --sharded-msm-log-n 0 --sharded-prove-log-n 0 --steps 20 $extra "$@" > gpurun_out/abf_$v$r.log 2>&1 || exit 1
    python - <<PY
import json
l = json.loads(open("gpurun_out/abf_$v$r.log").read().strip().splitlines()[-1])
print("$v$r [$extra]", "proofs/s (host inputs, 196 B) %.2f" % l["value"], "HBM-resident %.2f" % l["value_hbm_resident_inputs"], "latency %.2f" % l["single_proof_latency_ms"], "computeH solo %.3f" % l["roofline_ntt"]["compute_h_solo_ms"],
      "accum launch %.3f" % l["roofline"]["launch_ms"], flush=True)
PY
  done
done
