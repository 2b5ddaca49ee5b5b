#!/bin/bash
# The round's final measurement set (run on the GPU box through gpurun; tools/collect_profiles.py turns the outputs into profiles/).
#   part 1: bench line, kernel traces (default / one proof in flight), PMC traffic passes      part 2: N = 2^26 line
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
if [ "$1" = 1 ]; then
  rm -rf $O/r2_prof_def $O/r2_prof_if1b $O/r2_pmc_fetch2 $O/r2_pmc_write2
  timeout -k 10 400 python3 bench.py > $O/r2_bench_final.log 2>&1
  timeout -k 10 400 rocprofv3 --kernel-trace -d $O/r2_prof_def -o d -- python3 bench.py --sharded-msm-log-n 0 > $O/r2_prof_def.log 2>&1
  timeout -k 10 300 rocprofv3 --kernel-trace -d $O/r2_prof_if1b -o i -- python3 bench.py --in-flight 1 --steps 10 --sharded-msm-log-n 0 > $O/r2_prof_if1b.log 2>&1
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $O/r2_pmc_fetch2 -o f -- python3 bench.py --in-flight 1 --steps 2 --warmup 1 --no-cpu-baseline --no-host-inputs --sharded-msm-log-n 0 > $O/r2_pmc_fetch2.log 2>&1
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $O/r2_pmc_write2 -o w -- python3 bench.py --in-flight 1 --steps 2 --warmup 1 --no-cpu-baseline --no-host-inputs --sharded-msm-log-n 0 > $O/r2_pmc_write2.log 2>&1
  tail -c 300 $O/r2_bench_final.log
else
  timeout -k 10 600 python3 bench.py --log-n 26 --no-cpu-baseline --steps 8 --warmup 1 > $O/r2_bench26_final.log 2>&1
  tail -c 300 $O/r2_bench26_final.log
fi
