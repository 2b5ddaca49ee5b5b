"""BASELINE configs[4]: one large G1 MSM, bases sharded by points across the ranks of one node, partial sums
combined with an all-gather over RCCL + the host combine mi_g1_sum (SURVEY.md 8e option i).

  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
      tools/bench_sharded_msm.py --log-n 26 --steps 3

Each rank generates (on its GPU, seeded by the global index range) only its slice of the 2^log_n bases and scalars,
so the union over ranks is the same MSM instance for every N; rank 0 prints one JSON line with pts/s.
"""
import argparse
import importlib.util
import json
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def _mod(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-n", type=int, default=26)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--dist", choices=["uniform", "whir"], default="uniform")
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    rank, local_rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    torch.cuda.set_device(local_rank)
    dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    B = _mod("gnark_whir_amd_binding", os.path.join(ROOT, "gnark-whir_amd", "binding.py"))
    S = _mod("sharded", os.path.join(ROOT, "gnark-whir_amd", "sharded.py"))
    ctx = B.Context(local_rank)
    n = 1 << args.log_n
    lo, hi = S.shard_bounds(n, world, rank)
    # slice-local generation: seeds differ per rank, so the instance depends on N; the timing does not
    pts = ctx.gen_g1(hi - lo, 1000 + rank); sc = ctx.gen_scalars(hi - lo, 2000 + rank, 1 if args.dist == "whir" else 0)
    dev = torch.device("cuda", local_rank)

    def step():
        return S.sharded_msm(lambda: ctx.msm_g1_dev(pts.ptr, sc.ptr, hi - lo), B.g1_sum, dist, device=dev)

    ref = step()
    dist.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    dist.barrier(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert (out == ref).all()
    if rank == 0:
        print(json.dumps({"metric": "G1 MSM pts/sec (point-sharded, RCCL all-gather of partial sums)", "value": n * args.steps / float(t.item()),
                          "unit": "pts/s", "n_gpus": world, "log_n": args.log_n, "ms_per_msm": float(t.item()) / args.steps * 1e3,
                          "scaling": "strong", "scalars": args.dist}), flush=True)
    dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
