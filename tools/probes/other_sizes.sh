# the same step at other FFT domains, for the BASELINE mix and the census mix: tools/probes/other_sizes.sh [dist ...]   (default: whir census)
for d in ${@:-whir census}; do
for ln in 20 22 24 25; do
  timeout -k 10 500 python bench.py --log-n $ln --dist $d --no-cpu-baseline --sharded-msm-log-n 0 --sharded-prove-log-n 0 --no-sensitivity --no-live-pmc --steps $([ $ln -ge 24 ] && echo 12 || echo 40) > gpurun_out/size_${d}_$ln.log 2>&1 || { tail -3 gpurun_out/size_${d}_$ln.log; exit 1; }
  python3 - <<PY
import json
l = json.loads([x for x in open("gpurun_out/size_${d}_$ln.log") if x.startswith("{")][-1])
print("$d N=2^$ln value %.2f proofs/s (%.2f ms/step)  hbm-resident %.2f  single proof %.2f ms  host-input single %.2f ms  computeH solo %.3f ms  HBM in use %.1f GB" % (l["value"], l["ms_per_step"], l["value_hbm_resident_inputs"], l["single_proof_latency_ms"], l["single_proof_latency_host_inputs_ms"], l["roofline_ntt"]["compute_h_solo_ms"], l["hbm_in_use_gb"]), flush=True)
PY
done
done
