"""The PyTorch custom-op layer over the C-ABI (vae_eeg_xai_amd/ops.py): torch.ops.veex.* must
give bit-identical results to the direct ctypes binding of the same symbols, refuse CPU tensors (there
is no CPU kernel), trace under FakeTensor, and veex::sample must be differentiable."""
import numpy as np
import pytest

from oracle import vae_oracle as vo

pytestmark = pytest.mark.gpu


def test_ops_are_bit_identical_to_the_direct_binding():
    import torch
    from vae_eeg_xai_amd.net import Net
    L, B = 28, 48
    p = vo.init_params(L, "vae", seed=3, jitter=0.1)
    x = torch.from_numpy(vo.synth_topomaps(B, 5)).cuda()
    eps = torch.from_numpy(np.random.default_rng(1).standard_normal((B, L)).astype(np.float32)).cuda()
    res = []
    for via_ops in (False, True):
        net = Net(L, "vae")
        net.via_ops = via_ops
        net.set_weights_dict(p)
        out = {}
        net.train_fwd_bwd(x, eps)
        out["grads"] = net.grads.clone()
        net.adam_step(1e-3)
        out["params"] = net.params.clone()
        zm, zlv, z = net.encode(x, eps)
        out["zm"], out["zlv"], out["z"] = zm.clone(), zlv.clone(), z.clone()
        recon, logits = net.decode(z, want_logits=True)
        out["recon"], out["logits"] = recon.clone(), logits.clone()
        r2, losses = net.forward(x, eps)
        out["fwd"], out["losses"] = r2.clone(), losses.clone()
        out["attr"] = net.attribute(x, [0, 5], grad_x_input=True).clone()
        torch.cuda.synchronize()
        res.append(out)
    for k in res[0]:
        assert torch.equal(res[0][k], res[1][k]), k


def test_ops_through_torch_ops_namespace_and_fake_tensors():
    import torch
    from torch._subclasses.fake_tensor import FakeTensorMode
    from vae_eeg_xai_amd.net import Net
    L, B = 8, 6
    net = Net(L, "vae")
    net.init_weights(seed=1)
    x = torch.from_numpy(vo.synth_topomaps(B, 5)).cuda()
    ws = net.workspace(B)
    zm, zlv, z = torch.ops.veex.encode(net._h, net.params, net.state, x, None, 1, 42, 0, ws, L, True)
    recon, logits = torch.ops.veex.decode(net._h, net.params, net.state, z, ws, False)
    assert recon.shape == (B, 40, 40, 1) and logits.numel() == 0
    maps = torch.ops.veex.attribute(net._h, net.params, net.state, x, [1], False, ws)
    assert maps.shape == (1, B, 40, 40)
    with pytest.raises(NotImplementedError):       # no CPU kernel is registered: the product has no CPU path
        torch.ops.veex.sample(torch.zeros(2, 3), torch.zeros(2, 3), torch.zeros(2, 3))
    with FakeTensorMode():
        fx = torch.empty(5, 40, 40, 1, device="cuda")
        fp = torch.empty(10, device="cuda")
        a, b, c = torch.ops.veex.encode(0, fp, fp, fx, None, 1, 42, 0, fp, 28, True)
        assert a.shape == b.shape == c.shape == (5, 28)
        assert torch.ops.veex.attribute(0, fp, fp, fx, [1, 2, 3], True, fp).shape == (3, 5, 40, 40)


def test_sample_op_is_differentiable():
    import torch
    from vae_eeg_xai_amd.models import sample
    g = torch.Generator().manual_seed(0)
    zm = torch.randn(7, 5, generator=g).cuda().requires_grad_(True)
    zlv = (0.3 * torch.randn(7, 5, generator=g)).cuda().requires_grad_(True)
    eps = torch.randn(7, 5, generator=g).cuda()
    z = sample(zm, zlv, epsilon=eps)
    w = torch.randn(7, 5, generator=g).cuda()
    (z * w).sum().backward()
    ref_z = zm.detach() + torch.exp(0.5 * zlv.detach()) * eps
    assert torch.allclose(z.detach(), ref_z, rtol=1e-6, atol=1e-6)
    assert torch.allclose(zm.grad, w, rtol=1e-6, atol=1e-6)
    assert torch.allclose(zlv.grad, w * 0.5 * torch.exp(0.5 * zlv.detach()) * eps, rtol=1e-5, atol=1e-6)
    assert torch.library.opcheck(torch.ops.veex.sample.default, (zm.detach(), zlv.detach(), eps),
                                 test_utils=("test_schema", "test_faketensor")) is not None
