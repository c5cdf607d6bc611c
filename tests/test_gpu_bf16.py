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


@pytest.mark.parametrize("variant,B", [("vae", 200), ("ae", 167), ("vae", 512)])
def test_halo_tile_kernel_matches_the_general_bf16_kernel(variant, B):
    """VEEX_HALO=1 (default) routes the 3x3 conv / deconv forwards and input gradients of the bf16 engine through
    halo_gemm_bf16.hip (input images staged in LDS once, all taps and phases out of LDS).  Same bf16 operands and
    fp32 accumulation as the general kernel, another summation order (tap outer, channel slice inner): activations
    and gradients agree to fp32 rounding of the sums -- except where a bf16 consumer copy rounds the other way, which
    is bounded by the bf16 tolerance of this file.  B = 167: a ragged last sample group."""
    import torch
    from util import LAYERS
    from vae_eeg_xai_amd import _lib
    from vae_eeg_xai_amd.net import Net
    L = 28
    p = vo.init_params(L, variant, seed=3, jitter=0.1)
    x = vo.synth_topomaps(B, 5)
    eps = np.random.default_rng(7).standard_normal((B, L)).astype(np.float32) if variant == "vae" else None
    res = []
    try:
        for flag in ("0", "1"):
            _lib.setenv("VEEX_HALO", flag)
            net = Net(L, variant, compute="bf16")
            net.set_weights_dict(p)
            net.workspace(B).fill_(float("nan"))
            net.train_fwd_bwd(x, eps)
            torch.cuda.synchronize()
            res.append((net, net.grads.clone(), {ln: net.ws_region(B, "act." + ln).clone() for ln in LAYERS},
                        net.ws_region(B, "dpre.enc.conv1").clone()))
    finally:
        _lib.setenv("VEEX_HALO", None)
    (n0, g0, a0, d0), (n1, g1, a1, d1) = res

    def rel(a, b):
        return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()
    # the first halo layer differs from the general kernel only by the order of its fp32 sums
    assert rel(a1["enc.conv2"], a0["enc.conv2"]) < 2e-6
    worst = 0.0
    for ln in LAYERS:
        assert bool(torch.isfinite(a1[ln]).all()), ln
        worst = max(worst, rel(a1[ln], a0[ln]))
        assert rel(a1[ln], a0[ln]) < 5e-3, (ln, rel(a1[ln], a0[ln]))
    # ten layers of bf16 copies that may round the other way: element-wise the last gradient of the chain is only
    # comparable in the mean (the launch-level test below is the strict one)
    assert ((d1 - d0).norm() / d0.norm()).item() < BF16_GRAD_TOL
    for name, ref in n0.named_views(g0):
        got = n1.view([s for s in n1.specs if s.name == name][0], g1)
        assert rel(got, ref) < BF16_GRAD_TOL, (name, rel(got, ref))
    assert rel(g1[-4:-1], g0[-4:-1]) < 1e-4
    print("halo vs general bf16: worst activation difference", worst)


# (mode, Hin, Cin, Hout, Cout, k, stride, pad_begin): the forward and the input gradient of the seven big layers
HALO_GEOMS = [
    ("conv2.fwd", 0, 20, 64, 10, 128, 3, 2, 0), ("conv2.dgrad", 1, 10, 128, 20, 64, 3, 2, 0),
    ("conv3.fwd", 0, 10, 128, 5, 256, 3, 2, 0), ("conv3.dgrad", 1, 5, 256, 10, 128, 3, 2, 0),
    ("deconv1.fwd", 1, 4, 256, 8, 256, 3, 2, 0), ("deconv1.dgrad", 0, 8, 256, 4, 256, 3, 2, 0),
    ("deconv2.fwd", 1, 8, 256, 8, 128, 3, 1, 1), ("deconv2.dgrad", 0, 8, 128, 8, 256, 3, 1, 1),
    ("deconv3.fwd", 1, 8, 128, 17, 128, 3, 2, 0), ("deconv3.dgrad", 0, 17, 128, 8, 128, 3, 2, 0),
    ("deconv4.fwd", 1, 17, 128, 19, 64, 3, 1, 0), ("deconv4.dgrad", 0, 19, 64, 17, 128, 3, 1, 0),
    ("deconv5.fwd", 1, 19, 64, 39, 64, 3, 2, 0), ("deconv5.dgrad", 0, 39, 64, 19, 64, 3, 2, 0),
]


@pytest.mark.parametrize("geom", HALO_GEOMS, ids=[g[0] for g in HALO_GEOMS])
@pytest.mark.parametrize("epi", [0, 1])
def test_halo_tile_launch_equals_the_general_launch(geom, epi):
    """One launch, same bf16 operands, both kernels: the fp32 results differ only by the order of the fp32 sums
    (<= 2e-6 of the tensor's max), the bf16 copies by at most one bf16 rounding of such a difference, the column
    partial sums of the backward epilogue add up to the same totals.  Batch 171: a ragged last sample group.
    Forced plans (bands, channel slices, two channel tiles) are covered as well as the planner's own choice."""
    import ctypes as C
    import torch
    from vae_eeg_xai_amd import _lib
    from vae_eeg_xai_amd._lib import GemmDesc, lib, check
    label, mode, Hin, Cin, Hout, Cout, k, stride, pb = geom
    N = 171
    g = torch.Generator(device="cuda").manual_seed(hash(label) % 1000)
    A = torch.randn(N, Hin, Hin, Cin, device="cuda", generator=g)
    Abf = A.to(torch.bfloat16)
    W = torch.randn(k * k, Cout, Cin, device="cuda", generator=g) * (2.0 / (k * k * Cin)) ** 0.5
    Wbf = W.to(torch.bfloat16)                      # packed [tap][Cout][Cin]
    bias = torch.randn(Cout, device="cuda", generator=g)
    aux = torch.randn(N, Hout, Hout, Cout, device="cuda", generator=g)
    esc = torch.rand(Cout, device="cuda", generator=g) + 0.5
    osc, osh = torch.rand(Cout, device="cuda", generator=g) + 0.5, torch.randn(Cout, device="cuda", generator=g)
    lib.veexi_debug_hg_plan.restype = C.c_int
    lib.veexi_debug_hg_plan.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]

    def run():
        out = torch.full((N, Hout, Hout, Cout), float("nan"), device="cuda")
        obf = torch.full((N, Hout, Hout, Cout), float("nan"), device="cuda", dtype=torch.bfloat16)
        d = GemmDesc()
        d.struct_size = C.sizeof(GemmDesc)
        d.inp, d.weight, d.out = A.data_ptr(), W.data_ptr(), out.data_ptr()
        d.weight_bf16, d.in_bf16, d.out_bf16 = Wbf.data_ptr(), Abf.data_ptr(), obf.data_ptr()
        d.N, d.Hin, d.Win, d.Cin, d.Hout, d.Wout, d.Cout = N, Hin, Hin, Cin, Hout, Hout, Cout
        d.kh = d.kw = k
        d.stride, d.pad_begin, d.mode, d.epilogue, d.nsplit, d.relu = stride, pb, mode, epi, 1, 1
        part = None
        if epi == 0:
            d.bias = bias.data_ptr()
            d.out_scale, d.out_shift, d.out_channels = osc.data_ptr(), osh.data_ptr(), Cout
        else:
            d.aux, d.e_scale, d.e_channels = aux.data_ptr(), esc.data_ptr(), Cout
            mt = lib.veex_gather_gemm_m_tiles(C.byref(d))
            part = torch.full((mt, 3, Cout), float("nan"), device="cuda")
            d.partials = part.data_ptr()
        plan = (C.c_longlong * 8)()
        took = lib.veexi_debug_hg_plan(C.byref(d), plan)
        check(lib.veex_gather_gemm(C.byref(d), _lib.current_stream()), label)
        torch.cuda.synchronize()
        return out, obf, part, (tuple(plan) if took == 1 else None)

    try:
        _lib.setenv("VEEX_HALO", "0")
        ref, rbf, rpart, plan0 = run()
        assert plan0 is None
        _lib.setenv("VEEX_HALO", None)
        forced = [dict(), dict(bands=2), dict(bands=3, nt=1), dict(G=2, ch=64), dict(G=1, nt=2, occ=2), dict(G=5)]
        ran = 0
        for f in forced:
            _lib.setenv("VEEX_HG_PLAN", _lib.hg_plan_code(**f) or None)
            out, obf, part, plan = run()
            if plan is None:
                continue                      # this forced plan does not fit the layer: the general kernel ran
            ran += 1
            scale = ref.abs().max().item()
            assert bool(torch.isfinite(out).all()), (label, f)
            assert (out - ref).abs().max().item() <= 2e-6 * scale, (label, f, plan, (out - ref).abs().max().item() / scale)
            dbf = (obf.float() - rbf.float()).abs()
            assert bool((dbf <= 2.0 ** -7 * rbf.float().abs() + 1e-6 * scale).all()), (label, f)
            assert (dbf > 0).float().mean().item() < 1e-3, (label, f)         # and almost every copy has the same bits
            if epi == 1:
                tot, rtot = part.sum(0), rpart.sum(0)
                assert (tot - rtot).abs().max().item() <= 2e-5 * rtot.abs().max().item(), (label, f)
        assert ran >= 1 or label in ("conv3.fwd", "deconv1.fwd", "deconv1.dgrad"), label
    finally:
        for kk in ("VEEX_HG_PLAN", "VEEX_HALO"):
            _lib.setenv(kk, None)
