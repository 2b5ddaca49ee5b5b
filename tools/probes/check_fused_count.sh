cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/chk && timeout -k 10 300 rocprofv3 --kernel-trace -d gpurun_out/chk -o c -- python3 tools/prof_proof.py 23 4 > gpurun_out/chk.log 2>&1
python3 tools/prof_proof.py --summary $(ls gpurun_out/chk/*/c_results.db gpurun_out/chk/c_results.db 2>/dev/null | head -1) 4 | grep -E "last_sub|k_msm2_count|k_scan|kernel"
