#!/bin/bash
# kernel trace of bench.py with ONE proof in flight + the dispatch timeline of its last 45 ms (one proof's critical path); run through gpurun
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
rm -rf $O/r3_trace1
timeout -k 10 300 rocprofv3 --kernel-trace -d $O/r3_trace1 -o t -- python3 bench.py --in-flight 1 --steps 4 --warmup 1 --no-cpu-baseline --sharded-msm-log-n 0 --sharded-prove-log-n 0 "$@" > $O/r3_trace1.log 2>&1
DB=$(find $O/r3_trace1 -name '*_results.db' | head -1)
python3 tools/rocpd_summary.py timeline $DB $O/r3_trace1_timeline.txt 400
python3 tools/rocpd_summary.py stats $DB $O/r3_trace1_stats.csv > /dev/null
tail -c 400 $O/r3_trace1.log
