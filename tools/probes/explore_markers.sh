cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/mk && timeout -k 10 300 rocprofv3 --marker-trace --kernel-trace -d gpurun_out/mk -o m -- python3 tools/prof_proof.py 16 2 ranges > gpurun_out/mk.log 2>&1
tail -2 gpurun_out/mk.log | cut -c1-200
python3 - <<'PY'
import glob, sqlite3
db = glob.glob("gpurun_out/mk/**/*_results.db", recursive=True)[0]
c = sqlite3.connect(db)
for (n,) in c.execute("select name from sqlite_master where type in ('table','view') order by 1"):
    try:
        k = c.execute(f"select count(*) from {n}").fetchone()[0]
    except Exception as e:
        k = str(e)[:40]
    print(n, k)
PY
