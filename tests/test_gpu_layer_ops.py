"""GPU: the differentiable layer-level custom ops (torch.ops.veex.conv2d / conv_transpose2d, SURVEY 8b "autograd via
register_autograd mapping fwd -> (dgrad, wgrad) symbols") against torch's own float64 CPU autograd of the same Keras
layer (explicit TF padding / crop, as oracle/torch_ref.py), for every Conv2D / Conv2DTranspose geometry of
models.py:219-298, and composed into a two-layer chain under torch.autograd.  Tolerance 1e-5 relative to max-abs."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RTOL = 1e-5
LAYERS = [  # kind, Hin, Cin, Cout, k, stride, padding  (models.py:219-235, 270-298)
    ("conv", 20, 64, 128, 3, 2, "same"),
    ("conv", 10, 128, 256, 3, 2, "same"),
    ("deconv", 4, 256, 256, 3, 2, "same"),
    ("deconv", 8, 256, 128, 3, 1, "same"),
    ("deconv", 8, 128, 128, 3, 2, "valid"),
    ("deconv", 17, 128, 64, 3, 1, "valid"),
    ("deconv", 19, 64, 64, 3, 2, "valid"),
]


def _rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def _ref_layer(kind, x, w, b, stride, padding, relu):
    """float64 torch restatement of the Keras layer on NHWC input (TF 'same' padding before/after; Conv2DTranspose as
    conv_transpose2d without padding + crop at the start offset)."""
    import torch
    import torch.nn.functional as F
    from oracle import vae_oracle as vo
    h = x.permute(0, 3, 1, 2)
    k = w.shape[0]
    if kind == "conv":
        if padding == "same":
            _, pb, pa = vo.same_pad(x.shape[1], k, stride)
            h = F.pad(h, (pb, pa, pb, pa))
        y = F.conv2d(h, w.permute(3, 2, 0, 1), b, stride=stride)
    else:
        out, pb, _ = vo.deconv_geometry(x.shape[1], k, stride, padding)
        y = F.conv_transpose2d(h, w.permute(3, 2, 0, 1), None, stride=stride)[:, :, pb:pb + out, pb:pb + out]
        y = y + b.view(1, -1, 1, 1)
    y = y.permute(0, 2, 3, 1)
    return torch.relu(y) if relu else y


@pytest.mark.parametrize("layer", LAYERS, ids=[f"{l[0]}{l[1]}x{l[2]}to{l[3]}s{l[5]}{l[6]}" for l in LAYERS])
@pytest.mark.parametrize("relu", [True, False])
def test_layer_op_forward_and_gradients_match_f64_autograd(layer, relu):
    import torch
    import vae_eeg_xai_amd.ops  # noqa: F401  (registers torch.ops.veex.*)
    kind, H, cin, cout, k, s, pad = layer
    N = 64
    g = torch.Generator().manual_seed(17)
    x = torch.rand(N, H, H, cin, generator=g) - 0.3
    w = (torch.rand(k, k, cin, cout, generator=g) - 0.5) * (2.0 / np.sqrt(k * k * cin)) if kind == "conv" else \
        (torch.rand(k, k, cout, cin, generator=g) - 0.5) * (2.0 / np.sqrt(k * k * cin))
    b = torch.rand(cout, generator=g) - 0.5
    op = torch.ops.veex.conv2d if kind == "conv" else torch.ops.veex.conv_transpose2d
    xd, wd, bd = (t.cuda().requires_grad_(True) for t in (x, w, b))
    y = op(xd, wd, bd, s, pad, relu)
    xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
    yr = _ref_layer(kind, xr, wr, br, s, pad, relu)
    assert tuple(y.shape) == tuple(yr.shape)
    assert _rel(y.detach().cpu(), yr.detach()) < RTOL
    dy = torch.rand(yr.shape, generator=g) - 0.5
    if relu:                       # the device's ReLU decisions (an fp32 / f64 disagreement sits within rounding of the kink)
        mask = (y.detach().cpu() > 0)
        ypre = _ref_layer(kind, xr, wr, br, s, pad, False)
        kinks = ypre.detach()[mask != (ypre.detach() > 0)]
        assert kinks.numel() == 0 or float(kinks.abs().max()) < 1e-5
        yr = ypre * mask
    y.backward(dy.cuda())
    yr.backward(dy.double())
    assert _rel(xd.grad.cpu(), xr.grad) < RTOL
    assert _rel(wd.grad.cpu(), wr.grad) < RTOL
    assert _rel(bd.grad.cpu(), br.grad) < RTOL


def test_layer_ops_compose_under_autograd_and_reject_cpu_tensors():
    """conv2d -> conv_transpose2d chain: gradients flow through both ops; a CPU tensor is refused (no CPU kernel)."""
    import torch
    import vae_eeg_xai_amd.ops  # noqa: F401
    g = torch.Generator().manual_seed(3)
    x = torch.rand(32, 20, 20, 64, generator=g)
    w1 = (torch.rand(3, 3, 64, 128, generator=g) - 0.5) * 0.1
    b1 = torch.zeros(128)
    w2 = (torch.rand(3, 3, 64, 128, generator=g) - 0.5) * 0.1      # Conv2DTranspose kernel (k,k,Cout,Cin)
    b2 = torch.zeros(64)
    dev = [t.cuda().requires_grad_(True) for t in (x, w1, b1, w2, b2)]
    ref = [t.double().requires_grad_(True) for t in (x, w1, b1, w2, b2)]
    out = torch.ops.veex.conv_transpose2d(torch.ops.veex.conv2d(dev[0], dev[1], dev[2], 2, "same", True), dev[3], dev[4], 2, "same", False)
    outr = _ref_layer("deconv", _ref_layer("conv", ref[0], ref[1], ref[2], 2, "same", True), ref[3], ref[4], 2, "same", False)
    assert tuple(out.shape) == (32, 20, 20, 64)
    (out * out).sum().backward()
    (outr * outr).sum().backward()
    for d, r in zip(dev, ref):
        assert _rel(d.grad.cpu(), r.grad) < 1e-4        # two layers deep, ReLU kinks not pinned
    with pytest.raises((NotImplementedError, RuntimeError)):
        torch.ops.veex.conv2d(x, w1, b1, 2, "same", True)
