#!/bin/bash
# which MSM streams share a hardware queue, and the single-proof latency, for several stream creation orders (MI_MSM_STREAM_ORDER)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for order in 012345 301245 031245 012435 021345 013245 501234; do
  rm -rf gpurun_out/qp
  MI_MSM_STREAM_ORDER=$order timeout -k 10 300 rocprofv3 --kernel-trace -d gpurun_out/qp -o p -- python3 tools/prof_proof.py 23 8 > gpurun_out/qp.log 2>&1
  python3 - "$order" <<'PY'
import sqlite3, sys, glob
db = glob.glob("gpurun_out/qp/**/*_results.db", recursive=True)[0]
c = sqlite3.connect(db)
rows = c.execute("select stream_id, queue_id, count(*) from kernels where name like 'k_msm_accum_affine%' group by stream_id, queue_id order by stream_id").fetchall()
lat = [l for l in open("gpurun_out/qp.log") if l.startswith("proof latencies")][-1].split(":")[1].split()
print("order", sys.argv[1], "accumulate kernels (stream, queue, launches):", rows, "latencies", lat[2:], flush=True)
PY
done
rm -rf gpurun_out/qp
for order in 012345 301245 031245 012435 021345 013245 501234; do
  MI_MSM_STREAM_ORDER=$order python3 tools/prof_proof.py 23 8 | tail -1 | sed "s/^/unprofiled $order /"
done
