"""Full-size parity of every BASELINE.json configuration against the float64 oracle.

configs[1]: batch 512, L=28, fp32 (1e-5) and bf16 (stated bf16 tolerance, against the fp32 path);
configs[2]: local batch 1024 of the 8-GPU run, latent sweep L in {2, 8, 16, 28, 64, 128};
configs[3]: attribution batches of 4096 maps (a 256-map slice is compared with the oracle);
configs[4]: the models_standard.py autoencoder at batch 512.

The reference run is ``oracle/torch_ref.py`` in float64 on the box's host cores (a few seconds per
case).  Its ReLU branches are the device's (tests/util.device_masks asserts that every disagreement
between the fp32 device and the float64 oracle sits at a pre-activation within fp32 rounding of 0,
where the sub-gradient is ambiguous); everything else -- forward, three losses, all 44 gradients --
is compared at the north-star tolerance of 1e-5 (relative to the tensor's max-abs).
"""
import numpy as np
import pytest

from oracle import torch_ref as tr, vae_oracle as vo
from util import RTOL, device_masks, rel_err

pytestmark = pytest.mark.gpu


def _case(variant, L, B, seed):
    p = vo.init_params(L, variant, seed=seed, jitter=0.1)
    x = vo.synth_topomaps(B, seed + 2)
    eps = np.random.default_rng(seed + 4).standard_normal((B, L)).astype(np.float32) if variant == "vae" else None
    return p, x, eps


def _check_step(variant, L, B, seed):
    import torch
    from vae_eeg_xai_amd.net import Net
    torch.set_num_threads(min(16, torch.get_num_threads()))
    p, x, eps = _case(variant, L, B, seed)
    net = Net(L, variant)
    net.set_weights_dict(p)
    losses = net.train_fwd_bwd(x, eps).cpu().numpy()
    torch.cuda.synchronize()
    pre = tr.pre_activations(p, x, eps, variant)
    masks = device_masks(net, B, pre)
    total, rec, kl, g = tr.grads(p, x, eps, variant, masks=masks)
    assert abs(losses[0] - total) < RTOL * abs(total), (losses, total)
    assert abs(losses[1] - rec) < RTOL * abs(rec)
    if variant == "vae":
        assert abs(losses[2] - kl) < RTOL * max(abs(kl), 1.0)
    got = net.get_weights_dict(net.grads)
    assert set(got) == set(g)
    worst = max((rel_err(v, g[k]), k) for k, v in got.items())
    assert worst[0] < RTOL, worst
    return worst


def test_config1_batch512_train_step_matches_f64_oracle():
    """configs[1] (batch 512, L=28, fp32): every gradient within 1e-5 of the float64 oracle."""
    print("worst gradient:", _check_step("vae", 28, 512, seed=31))


@pytest.mark.parametrize("L", [2, 8, 16, 28, 64, 128])
def test_config2_local_batch1024_latent_sweep_matches_f64_oracle(L):
    """configs[2]: the per-GPU work of the 8-GPU run (global batch 8192 = 8 x 1024) over the latent sweep."""
    print("worst gradient:", _check_step("vae", L, 1024, seed=40 + L))


def test_config4_autoencoder_batch512_matches_f64_oracle():
    print("worst gradient:", _check_step("ae", 28, 512, seed=33))


def test_config3_attribution_batch4096_slice_matches_f64_oracle():
    """configs[3]: one attribution batch of 4096 maps; maps 1000..1255 are compared with the float64
    oracle (the encoder treats samples independently), the whole batch for finiteness and for
    bit-identity of the slice with a stand-alone call on it."""
    import torch
    from vae_eeg_xai_amd.net import Net
    L, B, lo, n = 28, 4096, 1000, 256
    p, x, _ = _case("vae", L, B, seed=51)
    net = Net(L, "vae")
    net.set_weights_dict(p)
    units = [3, 27]
    full = net.attribute(x, units, grad_x_input=True)
    assert torch.isfinite(full).all()
    xs = x[lo:lo + n]
    part = net.attribute(xs, units, grad_x_input=True)
    got = part.cpu().numpy()
    c = vo.encoder_fwd(p, np.asarray(xs, dtype=np.float64), "vae")
    masks = device_masks(net, n, c, layers=["enc.conv1", "enc.conv2", "enc.conv3", "enc.dense"])
    ref = tr.saliency(p, xs, units, "vae", grad_x_input=True, masks=masks)[..., 0]
    assert rel_err(got, ref) < RTOL
    # the same maps inside the 4096 batch: other tiles, same per-sample arithmetic
    assert rel_err(full[:, lo:lo + n].cpu().numpy(), ref) < RTOL


def test_config1_bf16_batch512_tracks_fp32():
    """configs[1] bf16 leg at its full batch: activations 2e-2, gradients 5e-2, losses 5e-3 of the fp32
    path (tolerances of tests/test_gpu_bf16.py, unchanged)."""
    import torch
    from vae_eeg_xai_amd.net import Net
    from util import LAYERS
    L, B = 28, 512
    p, x, eps = _case("vae", L, B, seed=35)
    res = {}
    for compute in ("f32", "bf16"):
        net = Net(L, "vae", compute=compute)
        net.set_weights_dict(p)
        losses = net.train_fwd_bwd(x, eps).cpu().numpy()
        torch.cuda.synchronize()
        res[compute] = (losses, {ln: net.ws_region(B, "act." + ln).cpu().numpy() for ln in LAYERS},
                        net.get_weights_dict(net.grads))
    l32, a32, g32 = res["f32"]
    l16, a16, g16 = res["bf16"]
    assert np.all(np.abs(l16 - l32) <= 5e-3 * np.maximum(np.abs(l32), 1.0)), (l16, l32)
    for ln in LAYERS:
        assert rel_err(a16[ln], a32[ln]) < 2e-2, ln
    for k in g32:
        assert rel_err(g16[k], g32[k]) < 5e-2, (k, rel_err(g16[k], g32[k]))
    assert any(not np.array_equal(g16[k], g32[k]) for k in g32)
