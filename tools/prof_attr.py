"""Attribution only (gradient x input, batch 4096, top-1 unit then all 28): target program for rocprofv3."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vae_eeg_xai_amd.net import Net
from bench import synth
B = 4096
net = Net(28, "vae"); net.init_weights(seed=1)
x = torch.from_numpy(synth(B, seed=3)).cuda()
for units in ([3], list(range(28))):
    for _ in range(3):
        net.attribute(x, units, grad_x_input=True)
torch.cuda.synchronize()
