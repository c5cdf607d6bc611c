"""world_size-2 gloo test (CPU) of the data-parallel step: the only exchange of a training step is
one SUM all-reduce of the flat gradient buffer (+3 loss scalars in its tail), scaled by 1/world in
the optimizer.  The gradients here come from the oracle (no GPU in this container); the collective,
the sharding and the epsilon indexing are the product's (vae_eeg_xai_amd.dist)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _flat(g, names):
    return np.concatenate([g[k].ravel() for k in names])


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import vae_oracle as vo
    from vae_eeg_xai_amd import dist as vdist
    L, B = 4, 6
    p = vo.init_params(L, "vae", seed=2, jitter=0.1)
    x = vo.synth_topomaps(B, 9)
    names = vo.trainable_names(L, "vae")
    assert vdist.world_size() == world and vdist.rank() == rank
    # the shuffle seed of VAE.fit: rank 0's value on every rank
    assert vdist.broadcast_int(1234567890123 + rank) == 1234567890123
    lo, hi = vdist.shard_range(B, rank, world)
    # epsilon keyed by the global sample index: the shard draws exactly its slice of the global noise
    eps_full = vo.philox_normal(7, vdist.global_first_sample(0, 0, 1, B), B, L)
    eps_mine = vo.philox_normal(7, vdist.global_first_sample(0, rank, world, hi - lo), hi - lo, L)
    assert np.array_equal(eps_mine, eps_full[lo:hi])
    c = vo.forward(p, x[lo:hi], eps_mine, "vae")
    g = vo.backward(p, c, "vae")
    flat = torch.tensor(np.concatenate([_flat(g, names), [c["total"], c["rec_loss"], c["kl_loss"], 0.0]]))
    # the two-bucket exchange (dist.TwoBucketReducer): one contiguous run reduced on its own, the pieces either side of
    # it after it -- element for element the sums of the single all-reduce, bit for bit
    pieces = flat.clone()
    first, count = 1000, flat.numel() // 2
    vdist.allreduce_sum_(pieces[first:first + count])
    vdist.allreduce_sum_outside_(pieces, first, count)
    vdist.allreduce_sum_(flat)
    assert torch.equal(pieces, flat)
    flat /= world                                   # the 1/world the Adam kernel folds into its gradient scale
    if rank == 0:
        cf = vo.forward(p, x, eps_full, "vae")
        gf = vo.backward(p, cf, "vae")
        ref = np.concatenate([_flat(gf, names), [cf["total"], cf["rec_loss"], cf["kl_loss"], 0.0]])
        err = np.abs(flat.numpy() - ref).max() / np.abs(ref).max()
        with open(out_path, "w") as f:
            f.write(repr(float(err)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_allreduce_equals_full_batch(tmp_path):
    out = os.path.join(tmp_path, "err.txt")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    assert float(open(out).read()) < 1e-12


def test_shard_range_partitions():
    from vae_eeg_xai_amd import dist as vdist
    for n in (0, 1, 7, 512, 8192, 1_000_000):
        for world in (1, 2, 3, 8):
            spans = [vdist.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    assert vdist.world_size() == 1 and vdist.rank() == 0
