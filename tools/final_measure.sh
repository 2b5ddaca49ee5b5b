#!/bin/bash
# The round's final measurement set (run on the GPU box through gpurun; tools/collect_profiles.py turns the outputs into profiles/).
#   1: the default bench line          1b: kernel trace of the proofs of the same workload (sharded legs and CPU baseline off)
#   2: kernel trace with one proof in flight + PMC traffic passes (FETCH_SIZE / WRITE_SIZE, separate runs)
#   3: VALU instruction counts over proofs alone (tools/prof_proof.py)       4: N = 2^26 line with the CPU baseline (~2 min of oracle)
#   5: n_committed sensitivity (2^16 / 2^18 / 2^20) + the 2-rank rehearsal on one GPU
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
Q="--no-cpu-baseline --sharded-msm-log-n 0 --sharded-prove-log-n 0"
case "$1" in
1)  timeout -k 10 600 python3 bench.py > $O/r4_bench_final.log 2>&1; tail -c 300 $O/r4_bench_final.log ;;
1b) rm -rf $O/r4_prof_def
    timeout -k 10 400 rocprofv3 --kernel-trace -d $O/r4_prof_def -o d -- python3 bench.py $Q --no-hbm-resident > $O/r4_prof_def.log 2>&1; tail -c 200 $O/r4_prof_def.log ;;
2)  rm -rf $O/r4_prof_if1 $O/r4_pmc_fetch $O/r4_pmc_write
    timeout -k 10 300 rocprofv3 --kernel-trace -d $O/r4_prof_if1 -o i -- python3 bench.py --in-flight 1 --steps 10 $Q --no-hbm-resident > $O/r4_prof_if1.log 2>&1
    # (--n-committed 0: the per-launch averages of k_msm_accum_affine29 must be those of the proof's four MSMs, not mixed with the two small Pedersen launches)
    timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $O/r4_pmc_fetch -o f -- python3 bench.py --in-flight 1 --steps 2 --warmup 1 $Q --no-hbm-resident --n-committed 0 > $O/r4_pmc_fetch.log 2>&1
    timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $O/r4_pmc_write -o w -- python3 bench.py --in-flight 1 --steps 2 --warmup 1 $Q --no-hbm-resident --n-committed 0 > $O/r4_pmc_write.log 2>&1
    tail -c 200 $O/r4_pmc_write.log ;;
3)  rm -rf $O/r4_pmc_valu
    timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_VALU -d $O/r4_pmc_valu -o v -- python3 tools/prof_proof.py 23 4 > $O/r4_pmc_valu.log 2>&1; tail -1 $O/r4_pmc_valu.log | cut -c1-200 ;;
4)  timeout -k 10 900 python3 bench.py --log-n 26 --steps 6 --warmup 1 > $O/r4_bench26_final.log 2>&1; tail -c 300 $O/r4_bench26_final.log ;;
5)  for nc in 65536 262144 1048576; do
      timeout -k 10 300 python3 bench.py $Q --n-committed $nc > $O/r4_nc_$nc.log 2>&1; tail -c 100 $O/r4_nc_$nc.log; echo
    done
    timeout -k 10 300 python3 bench.py $Q --n-committed 0 > $O/r4_nc_0.log 2>&1
    timeout -k 10 400 python3 bench.py --gpus 2 --rehearse-on-one-gpu --log-n 20 --steps 10 --no-cpu-baseline --sharded-msm-log-n 22 --sharded-prove-log-n 20 > $O/r4_rehearse2.log 2>&1; tail -c 200 $O/r4_rehearse2.log ;;
esac
