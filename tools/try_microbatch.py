"""Experiment: one train step of batch B as ONE call, against the same B samples as K micro-batches of B/K on K
streams (each with its own Net = own workspace and side stream), whose launches overlap each other's prologues,
epilogues and launch floors.  Timing only (the K gradient buffers would have to be summed).
usage: python tools/try_microbatch.py [dtype] [B]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from vae_eeg_xai_amd.net import Net  # noqa: E402

dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
x = torch.rand(B, 40, 40, 1, device="cuda")


def run(K, steps=30, warm=8, graph=False):
    nets = [Net(28, "vae", compute=dtype) for _ in range(K)]
    for n in nets:
        n.init_weights(seed=1)
    streams = [torch.cuda.Stream() for _ in range(K)]
    b = B // K

    def step():
        cur = torch.cuda.current_stream()
        for k in range(K):
            streams[k].wait_stream(cur)
            with torch.cuda.stream(streams[k]):
                nets[k].train_fwd_bwd(x[k * b:(k + 1) * b], None, first_sample=k * b)
        for k in range(K):
            cur.wait_stream(streams[k])
        nets[0].adam_step(1e-5)

    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / steps


for K in (1, 2, 4):
    print(f"{dtype} batch {B} as {K} micro-batch(es) on {K} stream(s): {run(K):.3f} ms/step", flush=True)
