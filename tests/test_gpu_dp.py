"""Data parallelism of the PRODUCT path (VAE._step / VAE.fit) on the GPU.

(1) one-process emulation: the "rank 0" and "rank 1" shares of one global batch go through the same
    VAE._shard_fwd_bwd the distributed step runs; epsilon must be bitwise the full batch's, the summed
    gradients the full batch's up to the order of fp32 sums -- equal and ragged shards, and a rank
    without samples;
(2) two real ranks (torch.distributed.run, gloo, both on the one GPU) run VAE.fit and must reproduce the
    one-process fit: same history, same weights."""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import vae_oracle as vo

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _vae(L, p):
    from vae_eeg_xai_amd.models import Encoder, Decoder, VAE, Adam
    v = VAE(Encoder(L), Decoder())
    v.compile(Adam(1e-5))
    v._net.set_weights_dict(p)
    return v


@pytest.mark.parametrize("n_global,world", [(64, 2), (37, 2), (5, 4), (3, 4)])
def test_shards_of_a_global_batch_sum_to_the_full_batch_step(n_global, world):
    import torch
    L = 28
    p = vo.init_params(L, "vae", seed=6, jitter=0.1)
    x = torch.from_numpy(vo.synth_topomaps(n_global + 11, 8)).cuda()
    index = torch.from_numpy(np.random.default_rng(0).permutation(n_global + 11)[:n_global]).cuda()
    full = _vae(L, p)
    full._net.samples_seen = 1000                       # some earlier global steps
    full._shard_fwd_bwd(x, None, index, 1, 0)
    eps_full = full._net.ws_region(n_global, "eps").clone().view(n_global, L)
    g_full = full._net.grads.clone()
    total = torch.zeros_like(g_full)
    from vae_eeg_xai_amd import dist as vdist
    for r in range(world):
        rep = _vae(L, p)
        rep._net.samples_seen = 1000
        rep._shard_fwd_bwd(x, None, index, world, r)
        lo, hi = vdist.shard_range(n_global, r, world)
        if hi > lo:     # the noise a sample sees does not depend on the number of ranks: bitwise
            eps = rep._net.ws_region(hi - lo, "eps").view(hi - lo, L)
            assert torch.equal(eps, eps_full[lo:hi]), (r, lo, hi)
        assert rep._net.samples_seen == 1000 + n_global
        total += rep._net.grads                         # the SUM all-reduce
    total /= world                                      # Adam's gscale = 1/world
    n = full._net.n_params
    for name, ref in full._net.named_views(g_full):
        spec = [s for s in full._net.specs if s.name == name][0]
        got = full._net.view(spec, total)
        err = (got - ref).abs().max().item() / (ref.abs().max().item() + 1e-30)
        assert err < 2e-5, (name, err)                  # order of fp32 sums over the batch only
    assert torch.allclose(total[n:n + 3], g_full[n:n + 3], rtol=2e-6, atol=0)   # the three loss scalars


@pytest.mark.parametrize("lr", [0.0, 1e-4])
def test_two_rank_fit_reproduces_the_one_process_fit(tmp_path, lr):
    """lr = 0: the weights stand still, so every logged metric of both epochs (training and validation, ragged
    batches included) must agree to the order of fp32 sums -- epsilon indexing, sharding, loss weighting and the
    metric means are all in it.  lr = 1e-4: the optimizer runs; Adam turns a rounding-level gradient into a step of
    ~lr whatever its size (sign-like in its first steps), so weights are compared to a fraction of the steps taken
    and the metrics relative to the total loss."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import dp_worker
    n, batch, epochs, L, seed, n_val = 101, 32, 2, 8, 5, 21     # ragged last batch (5 -> 3 + 2), ragged validation
    out = os.path.join(tmp_path, "dp2.npz")
    env = dict(os.environ, VEEX_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    port = 29600 + os.getpid() % 1000 + (7 if lr else 0)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "dp_worker.py"), out,
           *map(str, (n, batch, epochs, L, seed, n_val, lr))]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    dp = np.load(out)
    one = dp_worker.run_fit(n, batch, epochs, L, seed, n_val, lr)
    assert int(dp["samples_seen"]) == int(one["samples_seen"]) == epochs * (n + n_val)
    total = np.abs(one["h.loss"]).max()
    for k in one:
        if k.startswith("h."):
            if lr == 0.0:
                assert np.allclose(dp[k], one[k], rtol=1e-5, atol=0), (k, dp[k], one[k])
            else:
                assert np.abs(dp[k] - one[k]).max() < 2e-5 * total, (k, dp[k], one[k])
    steps = epochs * 4
    for k in one:
        if k.startswith("w."):
            d = np.abs(dp[k] - one[k]).max()
            assert d <= 0.5 * steps * lr + 1e-6 * np.abs(one[k]).max() * (lr > 0), (k, d)


def test_rccl_backend_single_rank_fit_equals_the_plain_fit(tmp_path):
    """The RCCL code path itself (backend "nccl": device tensors straight into the collectives, no host staging) on the
    one GPU of the box: a process group of ONE rank runs the data-parallel `VAE.fit` -- communicator set-up, the
    all-reduce of the flat gradient buffer, the seed broadcast and the metrics all-reduce all execute on RCCL -- and
    must reproduce the plain single-process fit (world = 1: same shards, same divisor)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import dp_worker
    n, batch, epochs, L, seed, n_val, lr = 70, 32, 2, 8, 9, 11, 1e-4
    out = os.path.join(tmp_path, "rccl1.npz")
    env = dict(os.environ, VEEX_DIST_BACKEND="nccl", MASTER_ADDR="127.0.0.1")
    port = 29700 + os.getpid() % 1000
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "dp_worker.py"), out,
           *map(str, (n, batch, epochs, L, seed, n_val, lr))]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    dp = np.load(out)
    one = dp_worker.run_fit(n, batch, epochs, L, seed, n_val, lr)
    for k in one:
        if k.startswith("h."):
            assert np.allclose(dp[k], one[k], rtol=1e-6, atol=0), (k, dp[k], one[k])
        elif k.startswith("w."):
            assert np.abs(dp[k] - one[k]).max() <= 1e-6 * max(np.abs(one[k]).max(), 1e-3), k


def test_bucket_event_changes_no_gradient_bit():
    """veex_net_set_bucket_event: the early column-sum flush and the fork that records the event reorder nothing that
    feeds a sum -- every gradient has the same bits with and without it; the event completes; the early bucket is the
    decoder's kernels and biases."""
    import torch
    from vae_eeg_xai_amd.net import Net
    L, B = 28, 96
    p = vo.init_params(L, "vae", seed=6, jitter=0.1)
    x = torch.from_numpy(vo.synth_topomaps(B, 8)).cuda()
    eps = torch.randn(B, L, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    res = []
    for variant in ("vae", "ae"):
        for overlap in (True, False):
            for use_event in (False, True):
                net = Net(L, variant, overlap=overlap)
                net.set_weights_dict({k: v for k, v in vo.init_params(L, variant, seed=6, jitter=0.1).items()})
                ev = torch.cuda.Event()
                if use_event:
                    net.set_bucket_event(ev)
                net.train_fwd_bwd(x, eps if variant == "vae" else None)
                if use_event:
                    ev.synchronize()
                torch.cuda.synchronize()
                res.append(net.grads.clone())
                first, count = net.grad_bucket()
                names = [s.name for s in net.specs if s.trainable and first <= s.offset < first + count]
                assert names[0] == "dec.dense.kernel" and names[-1] == "dec.deconv6.bias" and len(names) == 14
            assert torch.equal(res[-1], res[-2]), (variant, overlap)


def test_two_bucket_exchange_equals_the_single_all_reduce(tmp_path):
    """Two real ranks (gloo, one GPU): `VAE.fit` with the overlapped two-bucket gradient exchange (default) against
    the single all-reduce (VEEX_GRAD_BUCKETS=1) -- history and every weight BITWISE equal (the same element-wise
    sums over the same ranks), ragged batches and an empty-shard step included."""
    n, batch, epochs, L, seed, n_val, lr = 67, 32, 2, 8, 5, 9, 1e-3     # last batch 3 samples -> shards 2 + 1
    outs = []
    for buckets in ("1", "2"):
        out = os.path.join(tmp_path, f"dp_b{buckets}.npz")
        env = dict(os.environ, VEEX_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1", VEEX_GRAD_BUCKETS=buckets)
        port = 29800 + os.getpid() % 1000 + int(buckets)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
               "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "dp_worker.py"), out,
               *map(str, (n, batch, epochs, L, seed, n_val, lr))]
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(np.load(out))
    a, b = outs
    assert set(a.files) == set(b.files)
    for k in a.files:
        assert np.array_equal(a[k], b[k]), k
