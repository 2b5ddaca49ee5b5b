for r in 1 2 3 4; do
  for v in old p0 p1 p2; do
    unset MI355X_GROTH16_LIB; extra=""
    if [ $v = old ]; then export MI355X_GROTH16_LIB=$PWD/gnark-whir_amd/build_ab/libab.so; else extra="--stream-plan ${v#p}"; fi
    timeout -k 10 200 python bench.py --no-cpu-baseline --sharded-msm-log-n 0 --sharded-prove-log-n 0 --no-sensitivity --no-solo-legs --steps 30 $extra > gpurun_out/abp.log 2>&1 || { tail -3 gpurun_out/abp.log; exit 1; }
    python3 - <<PY
import json
l = json.loads([x for x in open("gpurun_out/abp.log") if x.startswith("{")][-1])
print("r$r $v", "value %.2f" % l["value"], "hbm %.2f" % l["value_hbm_resident_inputs"], "lat %.2f" % l["single_proof_latency_ms"], flush=True)
PY
  done
done
