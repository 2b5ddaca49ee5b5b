#!/bin/bash
# The round's final measurement set (run on the GPU box through gpurun; tools/collect_profiles.py turns the outputs into profiles/).
#   1: the default bench line          1b: kernel trace of the proofs of the same workload (sharded legs and CPU baseline off)
#   2: kernel trace with one proof in flight + PMC traffic passes (FETCH_SIZE / WRITE_SIZE, separate runs)
#   3: VALU instruction counts over proofs alone (tools/prof_proof.py)       4: N = 2^26 line with the CPU baseline (~2 min of oracle)
#   5: the 2-rank rehearsal on one GPU (the witness-distribution sensitivity is part of the default line since round 5)
#   6: PMC traffic of the SOLO Z-shaped level-1 launch (the basis of the roofline line) + its plain run
#   7: per-proof kernel profiles (census mix, BASELINE mix)
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
Q="--no-cpu-baseline --no-live-pmc --sharded-msm-log-n 0 --sharded-prove-log-n 0"
case "$1" in
1)  timeout -k 10 600 python3 bench.py > $O/r6_bench_final.log 2>&1; tail -c 300 $O/r6_bench_final.log ;;
1b) rm -rf $O/r6_prof_def
    timeout -k 10 400 rocprofv3 --kernel-trace -d $O/r6_prof_def -o d -- python3 bench.py $Q --no-hbm-resident --no-sensitivity > $O/r6_prof_def.log 2>&1; tail -c 200 $O/r6_prof_def.log ;;
2)  rm -rf $O/r6_prof_if1 $O/r6_pmc_fetch $O/r6_pmc_write
    timeout -k 10 300 rocprofv3 --kernel-trace -d $O/r6_prof_if1 -o i -- python3 bench.py --in-flight 1 --steps 10 $Q --no-hbm-resident --no-sensitivity > $O/r6_prof_if1.log 2>&1
    # (--n-committed 0: the per-launch averages of k_msm_accum_affine29 must be those of the proof's four MSMs, not mixed with the two small Pedersen launches)
    timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $O/r6_pmc_fetch -o f -- python3 bench.py --in-flight 1 --steps 2 --warmup 1 $Q --no-hbm-resident --no-sensitivity --no-solo-legs --n-committed 0 > $O/r6_pmc_fetch.log 2>&1
    timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $O/r6_pmc_write -o w -- python3 bench.py --in-flight 1 --steps 2 --warmup 1 $Q --no-hbm-resident --no-sensitivity --no-solo-legs --n-committed 0 > $O/r6_pmc_write.log 2>&1
    tail -c 200 $O/r6_pmc_write.log ;;
3)  rm -rf $O/r6_pmc_valu
    timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_VALU -d $O/r6_pmc_valu -o v -- python3 tools/prof_proof.py 23 4 > $O/r6_pmc_valu.log 2>&1; tail -1 $O/r6_pmc_valu.log | cut -c1-200 ;;
4)  timeout -k 10 900 python3 bench.py --log-n 26 --steps 6 --warmup 1 > $O/r6_bench26_final.log 2>&1; tail -c 300 $O/r6_bench26_final.log ;;
5)  timeout -k 10 400 python3 bench.py --gpus 2 --rehearse-on-one-gpu --log-n 20 --steps 10 --no-cpu-baseline --sharded-msm-log-n 22 --sharded-prove-log-n 20 > $O/r6_rehearse2.log 2>&1; tail -c 200 $O/r6_rehearse2.log ;;
6)  rm -rf $O/r6_pmc_solo_fetch $O/r6_pmc_solo_write
    timeout -k 10 200 python3 tools/solo_z_msm.py 23 3 > $O/r6_solo_z.log 2>&1
    timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE -d $O/r6_pmc_solo_fetch -o f -- python3 tools/solo_z_msm.py 23 2 > $O/r6_pmc_solo_fetch.log 2>&1
    timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE -d $O/r6_pmc_solo_write -o w -- python3 tools/solo_z_msm.py 23 2 > $O/r6_pmc_solo_write.log 2>&1
    rm -rf $O/r6_prof_solo_z
    timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/r6_prof_solo_z -o s -- python3 tools/solo_z_msm.py 23 5 > $O/r6_prof_solo_z.log 2>&1
    tail -2 $O/r6_solo_z.log ;;
7)  # kernel time per proof, one proof at a time, for the census mix and the BASELINE mix (tools/prof_proof.py's own summary)
    for m in census whir; do
      rm -rf $O/r6_prof_$m
      timeout -k 10 300 rocprofv3 --kernel-trace -d $O/r6_prof_$m -o p -- python3 tools/prof_proof.py 23 8 x $m > $O/r6_prof_$m.log 2>&1
      python3 tools/prof_proof.py --summary $(find $O/r6_prof_$m -name "*_results.db" | head -1) 8 > $O/r6_prof_${m}_summary.txt 2>&1
      rm -rf $O/r6_prof_$m
    done
    head -4 $O/r6_prof_census_summary.txt ;;
esac
