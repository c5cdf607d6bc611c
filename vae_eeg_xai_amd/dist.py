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


class TwoBucketReducer:
    """The gradient exchange of one step in two buckets, overlapped with the backward pass (the message
    is small, so every microsecond of it that is exposed counts).  ``net.train_fwd_bwd`` records an event where the
    decoder's kernel / bias gradients -- 56 % of the buffer, one contiguous range -- are final; their all-reduce is
    enqueued on a communication stream behind that event and runs while the encoder's backward pass still computes;
    the rest (two short pieces either side of the range, the three loss scalars in the tail) is reduced after the
    call; the optimizer waits for both.  Element for element the same sums as the single all-reduce (same ranks,
    same order per element), which stays available as ``VAE(..., grad_buckets=1)`` / ``VEEX_GRAD_BUCKETS=1``."""

    def __init__(self, net):
        self.net = net
        self.first, self.count = net.grad_bucket()
        self.comm = torch.cuda.Stream(device=net.device)
        self.event = torch.cuda.Event()
        net.set_bucket_event(self.event)

    def reduce_(self, ran_backward: bool = True):
        g = self.net.grads
        cur = torch.cuda.current_stream(self.net.device)
        early = g[self.first:self.first + self.count]
        if ran_backward:
            self.comm.wait_event(self.event)        # the bucket is final here, long before the call's last kernel
        else:
            self.comm.wait_stream(cur)              # (an empty shard: the buffer was zeroed on the current stream)
        with torch.cuda.stream(self.comm):
            _all_reduce_(early, dist.ReduceOp.SUM)
        allreduce_sum_outside_(g, self.first, self.count)
        cur.wait_stream(self.comm)
        return g

    def close(self):
        self.net.set_bucket_event(None)


def allreduce_sum_outside_(g: torch.Tensor, first: int, count: int) -> torch.Tensor:
    """SUM all-reduce of everything in the flat buffer EXCEPT ``g[first:first+count]`` (the late pieces of the
    two-bucket exchange: the tensors before the decoder's run, and the BatchNorm parameters + loss scalars after it)."""
    if world_size() > 1:
        if first > 0:
            _all_reduce_(g[:first], dist.ReduceOp.SUM)
        if first + count < g.numel():
            _all_reduce_(g[first + count:], dist.ReduceOp.SUM)
    return g


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
