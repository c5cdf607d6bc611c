"""Wide tiles of the lean fp32 kernel (VEEX_F32BM / VEEX_F32BN = 128) against the 64x64 tile: a train step and an
attribution pass.  The K order of every dot product is the same whatever the tile, so activations and input gradients
are bit-identical; the column sums behind bias / gamma / beta gradients add their per-tile partials in another
grouping, so those may differ by fp32 rounding of the sum."""
import sys, torch, numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from vae_eeg_xai_amd import _lib
from vae_eeg_xai_amd.net import Net
from oracle import vae_oracle as vo
L, B = 28, 96
p = vo.init_params(L, "vae", seed=3, jitter=0.1); x = vo.synth_topomaps(B, 5)
eps = np.random.default_rng(1).standard_normal((B, L)).astype(np.float32)
res = {}
for bm, bn in ((64, 64), (128, 64), (64, 128), (128, 128)):
    _lib.setenv("VEEX_F32BM", bm); _lib.setenv("VEEX_F32BN", bn)
    net = Net(L, "vae"); net.set_weights_dict(p); net.train_fwd_bwd(x, eps); torch.cuda.synchronize()
    res[(bm, bn)] = (net.grads.clone(), net.attribute(x, [1, 2]).clone(), net.ws_region(B, "dpre.enc.conv1").clone(), net)
g0, a0, d0, n0 = res[(64, 64)]
for k, (g, a, d, net) in res.items():
    worst = 0.0
    for name, ref in n0.named_views(g0):
        got = net.view([s for s in net.specs if s.name == name][0], g)
        worst = max(worst, ((got - ref).abs().max() / (ref.abs().max() + 1e-30)).item())
    print(k, "attribution identical:", torch.equal(a, a0), "dpre.conv1 identical:", torch.equal(d, d0), "worst grad rel diff: %.2e" % worst)
    assert torch.equal(a, a0) and torch.equal(d, d0) and worst < 2e-6
print("ok")
