"""Does a HIP graph of the whole train step help at the reference's batch sizes (32..128)?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vae_eeg_xai_amd.net import Net
from bench import synth

for B in (32, 64, 128, 512):
    net = Net(28, "vae"); net.init_weights(seed=1)
    x = torch.from_numpy(synth(B, seed=3)).cuda()
    eps = torch.randn(B, 28, device="cuda")
    def step():
        net.train_fwd_bwd(x, eps, accumulate_metrics=True)
        net.adam_step(1e-5)
    for _ in range(5): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100): step()
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / 100
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): step()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            eps.normal_()
            step()
    torch.cuda.synchronize()
    for _ in range(5): g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100): g.replay()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t0) / 100
    print(f"B={B:4d} eager {eager*1e3:.3f} ms/step ({B/eager:.0f}/s)   graph {graph*1e3:.3f} ms/step ({B/graph:.0f}/s)   loss {float(net.grads[net.n_params]):.3f}")
