"""VAE.fit throughput at the reference's own batch sizes (train.py:104-109: 32 / 64 / 128), full batches replayed from
a captured HIP graph (Net.train_step_graphed) against the eager fit.  One JSON line per (batch, mode).
usage: python tools/fit_throughput.py [dtype]"""
import json
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from vae_eeg_xai_amd.models import Encoder, Decoder, VAE, Adam  # noqa: E402

dtype = sys.argv[1] if len(sys.argv) > 1 else "f32"
x = np.random.default_rng(0).random((12800, 40, 40, 1), dtype=np.float32)
for bs in (8, 32, 64, 128, 256):
    for graph in (False, True):
        vae = VAE(Encoder(28), Decoder())
        vae._net.set_compute(dtype)
        vae.compile(Adam(1e-5))
        xd = torch.from_numpy(x).cuda()
        n = 12800 if bs >= 32 else 3200
        vae.fit(xd[:n], None, bs, 1, verbose=0, seed=1, graph=graph)     # warm-up epoch (capture included)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        vae.fit(xd[:n], None, bs, 2, verbose=0, seed=2, graph=graph)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(json.dumps({"metric": "VAE.fit samples/sec", "batch": bs, "graph": graph, "dtype": dtype, "value": 2 * n / dt,
                          "ms_per_step": 1e3 * dt / (2 * n / bs), "epochs": 2, "samples_per_epoch": n}), flush=True)
