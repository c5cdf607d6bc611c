"""GPU sanity sweep (not collected by pytest): config C3's latent sizes at local batch 1024, fp32 and bf16 -- one
train step each, loss finite and equal to the mean of the two half batches' losses (same epsilon)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import synth
from vae_eeg_xai_amd.net import Net

x = torch.from_numpy(synth(1024, seed=5)).cuda()
for dtype in ("f32", "bf16"):
    for L in (2, 8, 16, 28, 64, 128):
        net = Net(L, "vae", compute=dtype); net.init_weights(seed=1)
        eps = torch.randn(1024, L, device="cuda")
        full = net.train_fwd_bwd(x, eps).clone()
        g_full = net.grads[:net.n_params].clone()
        a = net.train_fwd_bwd(x[:512], eps[:512]).clone(); ga = net.grads[:net.n_params].clone()
        b = net.train_fwd_bwd(x[512:], eps[512:]).clone(); gb = net.grads[:net.n_params].clone()
        net.adam_step(1e-5)
        torch.cuda.synchronize()
        lerr = float(((a + b) / 2 - full).abs().max() / full.abs().max())
        gerr = float(((ga + gb) / 2 - g_full).abs().max() / g_full.abs().max())
        ok = bool(torch.isfinite(full).all()) and lerr < (1e-5 if dtype == "f32" else 1e-3) and gerr < (1e-4 if dtype == "f32" else 2e-2)
        print(f"{dtype} L={L:3d} loss={float(full[0]):.3f} split-loss err={lerr:.2e} split-grad err={gerr:.2e} {'ok' if ok else 'FAIL'}", flush=True)
        assert ok
    net = Net(28, "vae", compute=dtype); net.init_weights(seed=1)
    m = net.attribute(torch.cat([x] * 4), [0, 27], grad_x_input=True)
    assert bool(torch.isfinite(m).all()) and float(m.abs().max()) > 0
    print(dtype, "attribution at batch 4096 ok", flush=True)
