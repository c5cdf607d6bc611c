"""How much of a small-batch step is host enqueue time?  For batch B: wall time per train step (GPU-bound if the queue
stays full) and the host time to ENQUEUE a step (timed over 200 steps issued without synchronising, queue kept short
by a sync every 20).  usage: host_overhead.py [B ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vae_eeg_xai_amd.net import Net
for B in [int(a) for a in sys.argv[1:]] or [8, 64, 512]:
    net = Net(28, "vae"); net.init_weights(seed=1)
    x = torch.rand(B, 40, 40, 1, device="cuda")
    for i in range(10):
        net.train_fwd_bwd(x, None, first_sample=0); net.adam_step(1e-5)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(200):
        net.train_fwd_bwd(x, None, first_sample=0); net.adam_step(1e-5)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 200
    host = 0.0
    for blk in range(10):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(5):
            net.train_fwd_bwd(x, None, first_sample=0); net.adam_step(1e-5)
        host += time.perf_counter() - t0
    torch.cuda.synchronize()
    print(f"batch {B}: {wall * 1e3:.3f} ms/step wall, {host / 50 * 1e3:.3f} ms/step host enqueue (5 steps into an empty queue)", flush=True)
    # forward only (predict / serving)
    for i in range(10):
        net.forward(x, None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(200):
        net.forward(x, None)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 200
    host = 0.0
    for blk in range(10):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(5):
            net.forward(x, None)
        host += time.perf_counter() - t0
    torch.cuda.synchronize()
    print(f"batch {B}: forward {wall * 1e3:.3f} ms wall, {host / 50 * 1e3:.3f} ms host enqueue", flush=True)
    # the same forward replayed from a captured HIP graph (Net.forward_graphed; injected epsilon)
    eps = torch.randn(B, 28, device="cuda")
    for i in range(5):
        net.forward_graphed(x, eps)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(200):
        net.forward_graphed(x, eps)
    torch.cuda.synchronize()
    print(f"batch {B}: forward from a HIP graph {(time.perf_counter() - t0) / 200 * 1e3:.3f} ms wall", flush=True)
