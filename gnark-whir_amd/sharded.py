"""Point-sharded MSM across ranks (BASELINE configs[4], SURVEY.md 8e option i).

pk points are static, so rank k keeps slice k of the bases resident; per proof it receives slice k
of the scalars, runs the full single-GPU Pippenger on its slice (C-ABI mi_msm_g1_dev) and contributes
ONE normalised Jacobian partial (96 B for G1, 192 B for G2).  EC addition is not an RCCL reduction
op, so the "all-reduce of partial sums" is an all-gather of N x 96 bytes (RCCL when the tensors live
on the GPU, gloo on CPU) followed by the host combine mi_g1_sum / mi_g2_sum.  Latency-bound (~tens of
microseconds over xGMI); no bucket traffic crosses the links.
"""
from __future__ import annotations
import numpy as np


def shard_bounds(n: int, world: int, rank: int):
    """contiguous slices, the first (n % world) ranks take one extra pair"""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def all_gather_partials(partial: np.ndarray, dist, device=None) -> np.ndarray:
    """partial: (12,) or (24,) uint64 normalised Jacobian.  Returns (world, k) uint64."""
    import torch
    t = torch.from_numpy(partial.view(np.int64).copy())
    if device is not None:
        t = t.to(device)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return np.stack([o.cpu().numpy().view(np.uint64) for o in out])


def sharded_msm(local_msm, combine, dist, device=None) -> np.ndarray:
    """local_msm() -> this rank's partial; combine(parts) -> normalised sum (binding.g1_sum / g2_sum)."""
    return combine(all_gather_partials(local_msm(), dist, device))
