"""Interleaved A/B of an environment switch on attribution (batch 4096, all 28 units), one process.
usage: ab_attr.py VAR a b"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vae_eeg_xai_amd.net import Net
from vae_eeg_xai_amd import _lib
from bench import synth
var, vals = sys.argv[1], sys.argv[2:]
B = 4096
net = Net(28, "vae"); net.init_weights(seed=1)
x = torch.from_numpy(synth(B, seed=3)).cuda()
units = list(range(28))
for rnd in range(3):
    for v in vals:
        _lib.setenv(var, v)
        net.attribute(x, units, grad_x_input=True); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            net.attribute(x, units, grad_x_input=True)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        print(f"{var}={v}: {dt * 1e3:.2f} ms per 28 x {B} maps = {28 * B / dt / 1e6:.3f} M maps/s", flush=True)
