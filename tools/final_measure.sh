#!/bin/bash
# The round's final measurement set (run on the GPU box through gpurun; tools/collect_profiles.py turns the outputs into profiles/).
#   part 1: bench line + kernel trace of the default command      part 2: kernel trace with one proof in flight + PMC traffic passes
#   part 3: VALU instruction counts + instruction-rate and MFMA probes      part 4: N = 2^26 line (with the CPU baseline: ~2 min of oracle)
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
Q="--sharded-msm-log-n 0 --sharded-prove-log-n 0"
if [ "$1" = 1b ]; then
  rm -rf $O/r3_prof_def
  timeout -k 10 400 rocprofv3 --kernel-trace -d $O/r3_prof_def -o d -- python3 bench.py $Q --no-host-inputs > $O/r3_prof_def.log 2>&1
  tail -c 200 $O/r3_prof_def.log
elif [ "$1" = 1 ]; then
  rm -rf $O/r3_prof_def
  timeout -k 10 500 python3 bench.py > $O/r3_bench_final.log 2>&1
  # (no host-input leg under the profiler: its launches run beside PCIe waits, not beside two other proofs, and would pull the per-kernel
  #  averages away from what the timed region of the line reports)
  timeout -k 10 400 rocprofv3 --kernel-trace -d $O/r3_prof_def -o d -- python3 bench.py $Q --no-host-inputs > $O/r3_prof_def.log 2>&1
  tail -c 300 $O/r3_bench_final.log
elif [ "$1" = 2 ]; then
  rm -rf $O/r3_prof_if1 $O/r3_pmc_fetch $O/r3_pmc_write
  timeout -k 10 300 rocprofv3 --kernel-trace -d $O/r3_prof_if1 -o i -- python3 bench.py --in-flight 1 --steps 10 $Q > $O/r3_prof_if1.log 2>&1
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $O/r3_pmc_fetch -o f -- python3 bench.py --in-flight 1 --steps 2 --warmup 1 --no-cpu-baseline --no-host-inputs $Q > $O/r3_pmc_fetch.log 2>&1
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $O/r3_pmc_write -o w -- python3 bench.py --in-flight 1 --steps 2 --warmup 1 --no-cpu-baseline --no-host-inputs $Q > $O/r3_pmc_write.log 2>&1
  tail -c 200 $O/r3_pmc_write.log
elif [ "$1" = 3 ]; then
  rm -rf $O/r3_pmc_valu
  timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES -d $O/r3_pmc_valu -o v -- python3 bench.py --in-flight 1 --steps 2 --warmup 1 --no-cpu-baseline --no-host-inputs $Q > $O/r3_pmc_valu.log 2>&1
  (cd tools/bench_mfma && ./constmul) > $O/r3_probe_mfma_constmul.txt 2>&1
  cat $O/r3_probe_mfma_constmul.txt
else
  timeout -k 10 900 python3 bench.py --log-n 26 --steps 6 --warmup 1 > $O/r3_bench26_final.log 2>&1
  tail -c 300 $O/r3_bench26_final.log
fi
