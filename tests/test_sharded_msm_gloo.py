"""CPU suite, world_size 2 over gloo: the N>1 path of the point-sharded MSM with one rank per process (csrc/group.hip
mode 0: slice bounds, all-gather of the per-rank partial sums, combine by point additions through the C-ABI's mi_g1_sum /
mi_g2_sum).  There is no GPU in this container, so gloo stands in for RCCL's byte-typed all-gather and the oracle for the
per-rank MSM; on the GPU box tests/test_gpu_group.py runs the library's own sharded MSM and prove (2 and 3 ranks on device 0)
against the oracle."""
import importlib.util
import os
import sys
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _mod(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


def _worker(rank, world, port, n, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    for p in (os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    import cref
    binding = _mod("gnark_whir_amd_binding", os.path.join(ROOT, "gnark-whir_amd", "binding.py"))
    dist.init_process_group("gloo", rank=rank, world_size=world)

    def gather_and_sum(partial, combine):   # what combine_partials (csrc/group.hip) does with ncclAllGather(ncclUint8)
        t = torch.from_numpy(partial.view(np.int64).copy())
        out = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(out, t)
        return combine(np.stack([o.numpy().view(np.uint64) for o in out]))
    pts = cref.gen_g1(n, 123); sc = cref.gen_scalars(n, 456, 1)
    lo, hi = binding.shard_range(n, world, rank)
    got = gather_and_sum(cref.msm_g1(pts[lo:hi], sc[lo:hi]), binding.g1_sum)
    want = cref.msm_g1(pts, sc)
    p2 = cref.gen_g2(40, 7); s2 = cref.gen_scalars(40, 8, 0)
    lo2, hi2 = binding.shard_range(40, world, rank)
    got2 = gather_and_sum(cref.msm_g2(p2[lo2:hi2], s2[lo2:hi2]), binding.g2_sum)
    q.put((rank, bool(np.array_equal(got, want)), bool(np.array_equal(got2, cref.msm_g2(p2, s2))), (lo, hi)))
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [1001])
def test_sharded_msm_world2_gloo(n):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok1 and ok2 for _, ok1, ok2, _ in res)
    assert res[0][3] == (0, 500) and res[1][3] == (500, 1001)


def test_shard_bounds_cover_everything():
    binding = _mod("gnark_whir_amd_binding", os.path.join(ROOT, "gnark-whir_amd", "binding.py"))
    for n in (0, 1, 7, 8, 9, 1 << 20):
        for world in (1, 2, 3, 8):
            spans = [binding.shard_range(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1
