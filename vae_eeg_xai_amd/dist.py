"""Data parallelism over torch.distributed (backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the
CPU tests).  One process per GPU; the only collective of a training step is ONE all-reduce of the
flat gradient buffer (2 288 256 fp32 = 9.15 MB at L=28, plus the 3 loss scalars riding in its
tail); the 1/world factor is folded into the Adam kernel's gradient scale.
BatchNorm runs in inference mode, so there are no cross-sample statistics to sync."""
from __future__ import annotations

import torch
import torch.distributed as dist


def world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def shard_range(n: int, r: int, world: int):
    """Contiguous, near-equal split of n items: rank r gets [lo, hi)."""
    base, rem = divmod(n, world)
    lo = r * base + min(r, rem)
    return lo, lo + base + (1 if r < rem else 0)


def _all_reduce_(t: torch.Tensor, op) -> torch.Tensor:
    """dist.all_reduce in place; a device tensor under a host-only backend (gloo rehearsals of the
    N > 1 path on a one-GPU box) is staged through host memory."""
    if t.is_cuda and dist.get_backend() != "nccl":
        h = t.cpu()
        dist.all_reduce(h, op=op)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=op)
    return t


def allreduce_sum_(flat: torch.Tensor) -> torch.Tensor:
    """In-place SUM all-reduce of a flat fp32 buffer (no-op for a single process)."""
    if world_size() > 1:
        _all_reduce_(flat, dist.ReduceOp.SUM)
    return flat


def allreduce_max_(t: torch.Tensor) -> torch.Tensor:
    """In-place MAX all-reduce (the max-over-ranks wall time of bench.py)."""
    if world_size() > 1:
        _all_reduce_(t, dist.ReduceOp.MAX)
    return t


def broadcast_int(v: int, src: int = 0) -> int:
    """Rank ``src``'s integer on every rank (the shuffle seed of ``VAE.fit``)."""
    if world_size() == 1:
        return int(v)
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([int(v)], dtype=torch.int64, device=dev)
    dist.broadcast(t, src=src)
    return int(t.item())


def global_first_sample(step: int, r: int, world: int, local_batch: int) -> int:
    """Global index of a rank's first sample at a step: the Philox epsilon counter, so that the
    noise a sample sees does not depend on how many ranks share the batch."""
    return (step * world + r) * local_batch
