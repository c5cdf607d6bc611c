"""bf16-MFMA path (north_star: "fp32 reference plus bf16"): same C-ABI, compute='bf16'.
Stated tolerance (bf16 has an 8-bit significand, 2^-9 = 2e-3 per rounded operand): 2e-2 of the
tensor's max-abs for activations, 5e-2 for gradients (the loss gradient sigmoid(a) - t cancels, so
a 1e-3 error of a logit is a few 1e-2 of a small residual, and it is carried through ten layers),
5e-3 relative for the loss scalars."""
import numpy as np
import pytest

from oracle import vae_oracle as vo
from util import LAYERS, rel_err

pytestmark = pytest.mark.gpu
BF16_TOL = 2e-2
BF16_GRAD_TOL = 5e-2


def _pair(variant, L, B, seed=3):
    from vae_eeg_xai_amd.net import Net
    p = vo.init_params(L, variant, seed=seed, jitter=0.1)
    x = vo.synth_topomaps(B, seed + 2)
    eps = np.random.default_rng(seed + 4).standard_normal((B, L)).astype(np.float32) if variant == "vae" else None
    nets = []
    for compute in ("f32", "bf16"):
        net = Net(L, variant, compute=compute)
        net.set_weights_dict(p)
        nets.append(net)
    return p, x, eps, nets


def test_pack_weights_bf16_layouts():
    import ctypes as C
    import torch
    from vae_eeg_xai_amd import _lib
    src = torch.randn(9, 64, 128, device="cuda")
    for tr in (0, 1):
        dst = torch.empty(9 * 64 * 128, dtype=torch.bfloat16, device="cuda")
        _lib.check(_lib.lib.veex_pack_weights_bf16(_lib.ptr(src), _lib.ptr(dst), 9, 64, 128, tr, _lib.current_stream()))
        ref = src.transpose(1, 2).contiguous() if tr else src
        assert torch.equal(dst.view(ref.shape), ref.to(torch.bfloat16))   # RNE in both


@pytest.mark.parametrize("variant", ["vae", "ae"])
def test_bf16_weight_gradients_small_batch_vs_oracle(variant):
    """B=6: every gather GEMM is below the 128-row-tile threshold (fp32 kernels), the weight-gradient
    GEMMs run in bf16 -- checked directly against the float64 oracle."""
    import torch
    L, B = 28, 6
    p, x, eps, (_, net) = _pair(variant, L, B)
    c = vo.forward(p, x, eps, variant)
    losses = net.train_fwd_bwd(x, eps).cpu().numpy()
    torch.cuda.synchronize()
    assert abs(losses[0] - c["total"]) < 5e-3 * abs(c["total"])
    g = vo.backward(p, c, variant)
    got = net.get_weights_dict(net.grads)
    for k, v in got.items():
        assert rel_err(v, g[k]) < BF16_GRAD_TOL, (k, rel_err(v, g[k]))


@pytest.mark.parametrize("variant,B", [("vae", 192), ("ae", 200)])
def test_bf16_full_step_tracks_the_fp32_path(variant, B):
    """B >= 164: every big layer (forward, input gradient, weight gradient) takes the bf16 kernels;
    reference = the fp32 path of the same library (itself pinned to the oracle at 1e-5)."""
    import torch
    L = 28
    p, x, eps, (n32, n16) = _pair(variant, L, B)
    l32 = n32.train_fwd_bwd(x, eps).cpu().numpy()
    l16 = n16.train_fwd_bwd(x, eps).cpu().numpy()
    torch.cuda.synchronize()
    assert np.all(np.abs(l16 - l32) <= 5e-3 * np.maximum(np.abs(l32), 1.0)), (l16, l32)
    for ln in LAYERS:
        a32 = n32.ws_region(B, "act." + ln).cpu().numpy()
        a16 = n16.ws_region(B, "act." + ln).cpu().numpy()
        assert rel_err(a16, a32) < BF16_TOL, (ln, rel_err(a16, a32))
    g32, g16 = n32.get_weights_dict(n32.grads), n16.get_weights_dict(n16.grads)
    for k in g32:
        assert rel_err(g16[k], g32[k]) < BF16_GRAD_TOL, (k, rel_err(g16[k], g32[k]))
    print("worst bf16 gradient error:", max((rel_err(g16[k], g32[k]), k) for k in g32))
    # and the bf16 kernels really ran: results are not bit-identical to fp32
    assert any(not np.array_equal(g16[k], g32[k]) for k in g32)


def test_bf16_training_reduces_the_loss():
    import torch
    from vae_eeg_xai_amd.net import Net
    net = Net(28, "vae", compute="bf16")
    net.init_weights(seed=1)
    x = torch.from_numpy(vo.synth_topomaps(256, 5)).cuda()
    first = last = None
    for i in range(30):
        loss = float(net.train_fwd_bwd(x, None, first_sample=i * 256).cpu()[0])
        net.adam_step(1e-3)
        first = loss if first is None else first
        last = loss
    assert np.isfinite(last) and last < first


def test_lean_bf16_gather_kernel_matches_the_general_one():
    """VEEX_BFV3=1 routes every bf16 gather launch with a bf16 A copy through gather_gemm_bf16_v3.hip.  Same K order
    inside a dot product, so every activation and input gradient has the same bits; its tiles run over (image row,
    sample, column), so the per-tile column sums behind bias / gamma / beta gradients are grouped differently and
    may differ by fp32 rounding of the sum.  Off by default because it measures the same."""
    import torch
    from util import LAYERS
    from vae_eeg_xai_amd import _lib
    from vae_eeg_xai_amd.net import Net
    L, B = 28, 200
    p = vo.init_params(L, "vae", seed=3, jitter=0.1)
    x = vo.synth_topomaps(B, 5)
    eps = np.random.default_rng(7).standard_normal((B, L)).astype(np.float32)
    res = []
    try:
        for flag in ("0", "1"):
            _lib.setenv("VEEX_BFV3", flag)
            net = Net(L, "vae", compute="bf16")
            net.set_weights_dict(p)
            net.train_fwd_bwd(x, eps)
            torch.cuda.synchronize()
            res.append((net, net.grads.clone(), {ln: net.ws_region(B, "act." + ln).clone() for ln in LAYERS},
                        net.ws_region(B, "dpre.enc.conv1").clone()))
    finally:
        _lib.setenv("VEEX_BFV3", "0")
    (n0, g0, a0, d0), (n1, g1, a1, d1) = res
    for ln in LAYERS:
        assert torch.equal(a0[ln], a1[ln]), ln
    assert torch.equal(d0, d1)
    for name, ref in n0.named_views(g0):
        got = n1.view([s for s in n1.specs if s.name == name][0], g1)
        assert ((got - ref).abs().max() / (ref.abs().max() + 1e-30)).item() < 2e-6, name
