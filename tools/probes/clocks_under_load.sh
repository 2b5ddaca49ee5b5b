#!/bin/bash
# Samples the GPU's clocks and power (rocm-smi, read-only) while bench.py proves: is the job running at the 2.4 GHz the roofline figures assume?
python bench.py --no-cpu-baseline --sharded-msm-log-n 0 --sharded-prove-log-n 0 --no-sensitivity --no-live-pmc --no-solo-legs --steps 300 > gpurun_out/clk_bench.log 2>&1 &
BP=$!
sleep 8
for i in $(seq 1 16); do
  rocm-smi --showclocks --showpower --showtemp --csv 2>/dev/null | tr '\n' ' ' | cut -c1-600; echo
  sleep 0.5
done
wait $BP
python3 -c "
import json
l=json.loads([x for x in open('gpurun_out/clk_bench.log') if x.startswith('{')][-1]); print('value', l['value'], 'hbm', l['value_hbm_resident_inputs'])"
