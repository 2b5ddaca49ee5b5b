"""Which hardware queue does the HIP runtime give a stream?  Creates streams of given priorities in a given order, launches one small
kernel on each, and (under `rocprofv3 --kernel-trace`) the trace's queue_id per dispatch tells.  usage:
    rocprofv3 --kernel-trace -d DIR -o q -- python3 tools/probes/stream_queue_probe.py "H H H H H H N N N N N L L L L L"
    python3 tools/probes/stream_queue_probe.py --read DIR
Letters: H / N / L = create a stream of that priority; x = destroy the most recently created stream."""
import sys
if sys.argv[1] == "--read":
    import glob, sqlite3
    db = glob.glob(sys.argv[2] + "/**/*_results.db", recursive=True)[0]
    c = sqlite3.connect(db)
    rows = c.execute("select stream_id, queue_id, grid_x, min(start) from kernels group by stream_id, queue_id, grid_x order by 4").fetchall()
    for s, q, g, _ in rows:
        print(f"launch tag {g // 64:3d}: stream_id {s:3d} -> queue_id {q}")
    sys.exit(0)
import torch
plan = sys.argv[1].split()
prio = {"H": -1, "N": 0, "L": 1}
streams = []
for k, p in enumerate(plan):
    if p == "x":
        streams.pop()
        continue
    streams.append((k, p, torch.cuda.Stream(priority=prio[p])))
x = torch.zeros(1 << 16, device="cuda")
torch.cuda.synchronize()
for k, p, s in streams:
    with torch.cuda.stream(s):
        n = 64 * (k + 1)              # the grid size tags the launch in the trace
        y = x[:n * 4] + 1.0
torch.cuda.synchronize()
print(" ".join(f"{k}:{p}" for k, p, _ in streams))
