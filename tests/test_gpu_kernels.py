"""GPU: the lean fp32 gather kernel against the general one, launch by launch through the operator-level C-ABI
(veex_gather_gemm), at the full training batch -- where its tiles run over (image row, sample, column) or hold one
output pixel each, taps that are padding for a whole tile are skipped and the blocks are dealt per sample group over
the XCDs.  Skipped products are exact zeros and the k order is the same, so the outputs must be BIT-identical; the
per-tile column partials sit in other tiles, so their column totals are compared (fp32 rounding of a different
summation order).  Also: the BatchNorm-applied fp32 copy a producer writes (out_bn) against the affine prologue of
its consumer.  The library freezes its switches at first use: they are flipped with _lib.setenv."""
import ctypes as C
import os

import pytest

pytestmark = pytest.mark.gpu

CASES = [  # name, mode, N, Hin, Cin, Hout, Cout, k, stride, pad_begin, affine, epilogue
    ("deconv4 fwd", 1, 512, 17, 128, 19, 64, 3, 1, 0, 1, 0),
    ("deconv2 fwd", 1, 512, 8, 256, 8, 128, 3, 1, 1, 1, 0),
    ("deconv2 dgrad", 0, 512, 8, 128, 8, 256, 3, 1, 1, 0, 1),
    ("conv2 fwd", 0, 512, 20, 64, 10, 128, 3, 2, 0, 1, 0),
    ("deconv5 fwd (4 phases)", 1, 128, 19, 64, 39, 64, 3, 2, 0, 1, 0),
    ("conv3 dgrad (4 phases)", 1, 192, 5, 256, 10, 128, 3, 2, 0, 0, 1),
    ("deconv1 fwd (spatial affine)", 1, 512, 4, 256, 8, 256, 3, 2, 0, 2, 0),
    ("ragged batch", 1, 70, 17, 128, 19, 64, 3, 1, 0, 1, 0),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_lean_gather_kernel_is_bit_identical_to_the_general_one(case):
    import torch
    from vae_eeg_xai_amd import _lib
    from vae_eeg_xai_amd._lib import lib, check, GemmDesc
    name, mode, N, Hin, Cin, Hout, Cout, k, s, pb, aff, epi = case
    g = torch.Generator(device="cuda").manual_seed(5)

    def R(*shape):
        return torch.rand(*shape, device="cuda", generator=g) - 0.5

    x = R(N, Hin, Hin, Cin)
    W = R(k, k, Cin, Cout) if mode == 0 else R(k, k, Cout, Cin)
    bias = R(Cout)
    keep = []
    d = GemmDesc()
    d.inp, d.weight, d.bias = x.data_ptr(), W.data_ptr(), bias.data_ptr()
    if aff == 1:
        sc, sh = torch.rand(Cin, device="cuda", generator=g) + 0.5, R(Cin)
        keep += [sc, sh]
        d.in_scale, d.in_shift, d.affine_channels, d.affine_spatial = sc.data_ptr(), sh.data_ptr(), Cin, 0
    elif aff == 2:
        n = Hin * Hin * Cin
        sc, sh = torch.rand(n, device="cuda", generator=g) + 0.5, R(n)
        keep += [sc, sh]
        d.in_scale, d.in_shift, d.affine_channels, d.affine_spatial = sc.data_ptr(), sh.data_ptr(), n, 1
    d.N, d.Hin, d.Win, d.Cin, d.Hout, d.Wout, d.Cout = N, Hin, Hin, Cin, Hout, Hout, Cout
    d.kh = d.kw = k
    d.stride, d.pad_begin, d.mode = s, pb, mode
    d.epilogue, d.relu, d.nsplit = epi, 1, 1
    mt = 0
    if epi == 1:
        aux, esc = R(N, Hout, Hout, Cout), torch.rand(Cout, device="cuda", generator=g) + 0.5
        keep += [aux, esc]
        d.aux, d.e_scale, d.e_channels = aux.data_ptr(), esc.data_ptr(), Cout
        tmp = torch.empty(1, device="cuda")
        d.out = tmp.data_ptr()
        mt = lib.veex_gather_gemm_m_tiles(C.byref(d))
        assert mt > 0
    saved = {v: os.environ.get(v) for v in ("VEEX_GGV3",)}
    res = []
    try:
        for v3 in ("0", "1"):
            _lib.setenv("VEEX_GGV3", v3)
            out = torch.full((N, Hout, Hout, Cout), 7.0, device="cuda")
            d.out = out.data_ptr()
            part = None
            if epi == 1:
                part = torch.zeros(mt * 3 * Cout, device="cuda")
                d.partials = part.data_ptr()
            check(lib.veex_gather_gemm(C.byref(d), _lib.current_stream()), name)
            torch.cuda.synchronize()
            res.append((out, part))
    finally:
        for v, val in saved.items():
            _lib.setenv(v, val)
    ref = res[0][0]
    assert torch.isfinite(ref).all() and float(ref.abs().max()) > 0
    assert torch.equal(ref, res[1][0]), "lean kernel (its tile order: one output pixel per tile, centre-out, where it applies)"
    if epi == 1:
        ta = res[0][1].view(mt, 3, Cout).sum(0)
        for r in res[1:]:
            tc = r[1].view(mt, 3, Cout).sum(0)
            assert float((ta - tc).abs().max()) <= 1e-5 * float(ta.abs().max())


def test_batchnorm_applied_copy_equals_the_consumer_prologue():
    """VeexGemmDesc::out_bn: the producer's epilogue writes fma(out, scale, shift); a consumer that reads that copy
    without in_scale / in_shift gives the same bits as one that reads `out` with the affine prologue (padding
    included: zero stays zero either way), for the gather GEMM and for the weight-gradient GEMM; veex_conv1_fwd_copies
    writes the same copy for conv1."""
    import torch
    from vae_eeg_xai_amd import _lib
    from vae_eeg_xai_amd._lib import lib, check, GemmDesc, WgradDesc
    g = torch.Generator(device="cuda").manual_seed(9)

    def R(*shape):
        return torch.rand(*shape, device="cuda", generator=g) - 0.5

    st = _lib.current_stream
    N = 128
    # producer: deconv3-shaped (8 -> 17, 128 -> 128, stride 2, valid); consumer: deconv4-shaped (17 -> 19, 128 -> 64)
    x, W3, b3 = R(N, 8, 8, 128), R(3, 3, 128, 128) * 0.1, R(128)
    sc, sh = torch.rand(128, device="cuda", generator=g) + 0.5, R(128)
    a3, y3 = torch.empty(N, 17, 17, 128, device="cuda"), torch.empty(N, 17, 17, 128, device="cuda")
    d = GemmDesc()
    d.inp, d.weight, d.bias, d.out, d.out_bn = x.data_ptr(), W3.data_ptr(), b3.data_ptr(), a3.data_ptr(), y3.data_ptr()
    d.out_scale, d.out_shift, d.out_channels = sc.data_ptr(), sh.data_ptr(), 128
    d.N, d.Hin, d.Win, d.Cin, d.Hout, d.Wout, d.Cout = N, 8, 8, 128, 17, 17, 128
    d.kh = d.kw = 3
    d.stride, d.pad_begin, d.mode, d.epilogue, d.relu, d.nsplit = 2, 0, 1, 0, 1, 1
    check(lib.veex_gather_gemm(C.byref(d), st()), "producer")
    torch.cuda.synchronize()
    assert float(a3.max()) > 0 and float((a3 == 0).float().mean()) > 0.1          # ReLU is active
    assert torch.equal(y3, torch.addcmul(sh.expand_as(a3), a3, sc.expand_as(a3))) or \
        float((y3 - (a3 * sc + sh)).abs().max()) <= 1e-6                          # fma vs mul+add: one rounding apart
    W4, b4 = R(3, 3, 64, 128) * 0.1, R(64)
    outs = []
    for src, affine in ((a3, True), (y3, False)):
        o = torch.empty(N, 19, 19, 64, device="cuda")
        c = GemmDesc()
        c.inp, c.weight, c.bias, c.out = src.data_ptr(), W4.data_ptr(), b4.data_ptr(), o.data_ptr()
        if affine:
            c.in_scale, c.in_shift, c.affine_channels = sc.data_ptr(), sh.data_ptr(), 128
        c.N, c.Hin, c.Win, c.Cin, c.Hout, c.Wout, c.Cout = N, 17, 17, 128, 19, 19, 64
        c.kh = c.kw = 3
        c.stride, c.pad_begin, c.mode, c.epilogue, c.relu, c.nsplit = 1, 0, 1, 0, 1, 1
        check(lib.veex_gather_gemm(C.byref(c), st()), "consumer")
        torch.cuda.synchronize()
        outs.append(o)
    assert torch.equal(outs[0], outs[1])
    # weight gradient of the consumer: small = its input (affine or copy), big = d(pre-activation)
    dy = R(N, 19, 19, 64)
    grads = []
    for src, affine in ((a3, True), (y3, False)):
        w = WgradDesc()
        w.big, w.small = dy.data_ptr(), src.data_ptr()
        if affine:
            w.small_scale, w.small_shift, w.small_affine_channels = sc.data_ptr(), sh.data_ptr(), 128
        w.N, w.Hb, w.Wb, w.Cb, w.Hs, w.Ws, w.Cs = N, 19, 19, 64, 17, 17, 128
        w.kh = w.kw = 3
        w.stride, w.pad_begin = 1, 0
        ns, per = C.c_int32(), C.c_int32()
        check(lib.veex_wgrad_plan(C.byref(w), C.byref(ns), C.byref(per)), "plan")
        w.nsplit, w.rows_per_split = ns.value, per.value
        slab = torch.empty(ns.value * 9 * 64 * 128, device="cuda")
        w.out = slab.data_ptr()
        check(lib.veex_wgrad_gemm(C.byref(w), st()), "wgrad")
        torch.cuda.synchronize()
        grads.append(slab)
    assert torch.equal(grads[0], grads[1])
    # conv1: the copy next to the activation
    xi, W1, b1 = torch.rand(N, 40, 40, device="cuda", generator=g), R(3, 3, 1, 64), R(64)
    s1, h1 = torch.rand(64, device="cuda", generator=g) + 0.5, R(64)
    a1, y1, a1b = (torch.empty(N, 20, 20, 64, device="cuda") for _ in range(3))
    check(lib.veex_conv1_fwd_copies(xi.data_ptr(), W1.data_ptr(), b1.data_ptr(), a1.data_ptr(), s1.data_ptr(), h1.data_ptr(),
                                    y1.data_ptr(), None, N, st()), "conv1 copies")
    check(lib.veex_conv1_fwd(xi.data_ptr(), W1.data_ptr(), b1.data_ptr(), a1b.data_ptr(), N, st()), "conv1")
    torch.cuda.synchronize()
    assert torch.equal(a1, a1b)
    assert float((y1 - (a1 * s1 + h1)).abs().max()) <= 1e-6


SMALL = [  # name, mode, N, Hin, Cin, Hout, Cout, k, stride, pad_begin, epilogue   (few-tile launches: the 32x32 split-K kernel)
    ("deconv4 fwd", 1, 64, 17, 128, 19, 64, 3, 1, 0, 0),
    ("deconv2 fwd", 1, 5, 8, 256, 8, 128, 3, 1, 1, 0),
    ("deconv2 dgrad", 0, 64, 8, 128, 8, 256, 3, 1, 1, 1),
    ("conv2 fwd", 0, 33, 20, 64, 10, 128, 3, 2, 0, 0),
    ("deconv5 fwd (4 phases)", 1, 3, 19, 64, 39, 64, 3, 2, 0, 0),
    ("conv3 dgrad (4 phases)", 1, 17, 5, 256, 10, 128, 3, 2, 0, 1),
    ("deconv1 dgrad", 0, 64, 8, 256, 4, 256, 3, 2, 0, 1),
    ("one row", 1, 1, 4, 256, 8, 256, 3, 2, 0, 0),
]


@pytest.mark.parametrize("case", SMALL, ids=[c[0] for c in SMALL])
def test_small_launch_kernel_matches_the_64x64_kernel(case):
    """gather_gemm_v4 (32x32 tiles, K split over the four waves of a block, taken for launches of few tiles) against
    the lean 64x64 kernel on the same operands: outputs, the BatchNorm-applied copy and the column totals of the
    backward epilogue agree to fp32 rounding of a different summation order (1e-5 relative to max-abs; measured ~1e-6)."""
    import torch
    from vae_eeg_xai_amd import _lib
    from vae_eeg_xai_amd._lib import lib, check, GemmDesc
    name, mode, N, Hin, Cin, Hout, Cout, k, s, pb, epi = case
    g = torch.Generator(device="cuda").manual_seed(11)

    def R(*shape):
        return torch.rand(*shape, device="cuda", generator=g) - 0.5

    x = R(N, Hin, Hin, Cin)
    W = R(k, k, Cin, Cout) if mode == 0 else R(k, k, Cout, Cin)
    bias = R(Cout)
    osc, osh = torch.rand(Cout, device="cuda", generator=g) + 0.5, R(Cout)
    keep = [x, W, bias, osc, osh]
    d = GemmDesc()
    d.inp, d.weight, d.bias = x.data_ptr(), W.data_ptr(), bias.data_ptr()
    if mode == 0:
        wt = torch.empty(W.numel(), device="cuda")
        check(lib.veex_transpose_weights_f32(W.data_ptr(), wt.data_ptr(), k * k, Cin, Cout, _lib.current_stream()), "transpose")
        keep.append(wt)
        d.weight_t = wt.data_ptr()
    d.N, d.Hin, d.Win, d.Cin, d.Hout, d.Wout, d.Cout = N, Hin, Hin, Cin, Hout, Hout, Cout
    d.kh = d.kw = k
    d.stride, d.pad_begin, d.mode = s, pb, mode
    d.epilogue, d.relu, d.nsplit = epi, 1, 1
    if epi == 1:
        aux, esc = R(N, Hout, Hout, Cout), torch.rand(Cout, device="cuda", generator=g) + 0.5
        keep += [aux, esc]
        d.aux, d.e_scale, d.e_channels = aux.data_ptr(), esc.data_ptr(), Cout
    else:
        d.out_scale, d.out_shift, d.out_channels = osc.data_ptr(), osh.data_ptr(), Cout
    saved = os.environ.get("VEEX_GGV4")
    res = []
    try:
        for v4 in ("0", "1"):
            _lib.setenv("VEEX_GGV4", v4)
            out = torch.full((N, Hout, Hout, Cout), 7.0, device="cuda")
            obn = torch.full((N, Hout, Hout, Cout), 7.0, device="cuda")
            d.out = out.data_ptr()
            part, mt = None, 0
            if epi == 1:
                mt = lib.veex_gather_gemm_m_tiles(C.byref(d))
                assert mt > 0
                part = torch.zeros(mt * 3 * Cout, device="cuda")
                d.partials = part.data_ptr()
            else:
                d.out_bn = obn.data_ptr()
            check(lib.veex_gather_gemm(C.byref(d), _lib.current_stream()), name)
            torch.cuda.synchronize()
            res.append((out, obn, part, mt))
    finally:
        _lib.setenv("VEEX_GGV4", saved)
    (o0, b0, p0, m0), (o1, b1, p1, m1) = res
    assert torch.isfinite(o0).all() and float(o0.abs().max()) > 0
    scale = float(o0.abs().max())
    assert float((o0 - o1).abs().max()) <= 1e-5 * scale
    if epi == 0:
        assert float((b0 - b1).abs().max()) <= 1e-5 * float(b0.abs().max())
        assert not torch.equal(o0, torch.full_like(o0, 7.0))
    else:
        assert m1 >= m0                                    # 32-row tiles: at least as many partial rows
        t0, t1 = p0.view(m0, 3, Cout).sum(0), p1.view(m1, 3, Cout).sum(0)
        assert float((t0 - t1).abs().max()) <= 1e-5 * float(t0.abs().max())


@pytest.mark.parametrize("B", [64, 512])
def test_split_fp32_gather_kernel_equals_the_fp32_mfma_kernel(B):
    """VeexGemmDesc::weight_planes = 3 (gather_gemm_x3.hip): every forward / input-gradient launch of the train step
    on the bf16 matrix cores from three-term operand splits against the same launch on the fp32 MFMA -- an fp32 result:
    the two differ by accumulation order only (<= 5e-6 of the tensor's max-abs; a bf16 launch differs by ~4e-3)."""
    import ctypes as C
    import torch
    from vae_eeg_xai_amd import _lib, roofline as rl
    from vae_eeg_xai_amd._lib import GemmDesc, check, lib
    took = 0
    f32 = {label: fn for label, _, fn in rl.gemm_cases(B, False) if not label.endswith("wgrad")}
    keep32 = rl.gemm_cases._keep
    for label, _, fn in rl.gemm_cases(B, False, x3=True):
        if label.endswith("wgrad"):
            continue
        d3 = fn.__defaults__[0]
        d = GemmDesc.from_buffer_copy(d3)          # the same operands through the fp32-MFMA kernel
        d.weight_bf16, d.weight_planes = None, 0
        shape = (d.N, d.Hout, d.Wout, d.Cout)
        ref, out = torch.empty(shape, device="cuda"), torch.full(shape, 7.0, device="cuda")
        d.out, d3.out = ref.data_ptr(), out.data_ptr()
        part = part3 = None
        if d.epilogue == 1:
            rows = d.N * d.Hout * d.Wout // 32 + 8
            part, part3 = torch.zeros(rows * 3 * d.Cout, device="cuda"), torch.zeros(rows * 3 * d.Cout, device="cuda")
            d.partials, d3.partials = part.data_ptr(), part3.data_ptr()
        check(lib.veex_gather_gemm(C.byref(d), _lib.current_stream()), label)
        check(lib.veex_gather_gemm(C.byref(d3), _lib.current_stream()), label)
        torch.cuda.synchronize()
        scale = ref.abs().max().item()
        assert scale > 0 and bool(torch.isfinite(out).all())
        assert (out - ref).abs().max().item() <= 5e-6 * scale, label
        if d.epilogue == 1:     # column sums behind the bias / gamma / beta gradients (tile heights differ: compare totals)
            mt, mt3 = lib.veex_gather_gemm_m_tiles(C.byref(d)), lib.veex_gather_gemm_m_tiles(C.byref(d3))
            t, t3 = part[:mt * 3 * d.Cout].view(mt, 3, d.Cout).sum(0), part3[:mt3 * 3 * d.Cout].view(mt3, 3, d.Cout).sum(0)
            assert (t - t3).abs().max().item() <= 2e-5 * t.abs().max().item(), label
            took += mt3 != mt
    del f32, keep32
    assert B < 512 or took >= 3        # at batch 512 the split kernel's own (taller) tiles ran


@pytest.mark.gpu
@pytest.mark.parametrize("B", [64, 512])
def test_split_fp32_weight_gradient_kernel_equals_the_fp32_mfma_kernel(B):
    """VeexWgradDesc::bf16 = 2 (wgrad_gemm_x3.hip: row-major three-term planes, transposing LDS reads, six bf16 MFMAs per
    product) against the same launch on the fp32 MFMA, slab sums compared: an fp32 result (<= 5e-6 of the gradient's
    max-abs; a bf16 launch differs by ~4e-3).  Every weight-gradient launch of the train step, with and without the
    producer's BatchNorm on the operands."""
    import ctypes as C
    import torch
    from vae_eeg_xai_amd import _lib, roofline as rl
    from vae_eeg_xai_amd._lib import WgradDesc, check, lib
    n = 0
    for label, _, fn in rl.gemm_cases(B, False, x3=True):
        if not label.endswith("wgrad"):
            continue
        d3 = fn.__defaults__[0]
        assert d3.bf16 == 2
        d = WgradDesc.from_buffer_copy(d3)
        d.bf16 = 0
        M = d.kh * d.kw * d.Cb
        ref = torch.empty(d.nsplit * M * d.Cs, device="cuda")
        out = torch.full((d3.nsplit * M * d3.Cs,), 7.0, device="cuda")
        d.out, d3.out = ref.data_ptr(), out.data_ptr()
        check(lib.veex_wgrad_gemm(C.byref(d), _lib.current_stream()), label)
        check(lib.veex_wgrad_gemm(C.byref(d3), _lib.current_stream()), label)
        torch.cuda.synchronize()
        g, g3 = ref.view(d.nsplit, M, d.Cs).sum(0), out.view(d3.nsplit, M, d3.Cs).sum(0)
        scale = g.abs().max().item()
        assert scale > 0 and bool(torch.isfinite(g3).all()), label
        assert (g3 - g).abs().max().item() <= 5e-6 * scale, label
        n += 1
        # the producer's BatchNorm applied while staging: on the tap-shifted operand (padding stays 0), then on the other
        for side in ("big", "small"):
            ch = d.Cb if side == "big" else d.Cs
            sc, sh = torch.rand(ch, device="cuda") + 0.5, torch.randn(ch, device="cuda") * 0.1
            for dd in (d, d3):
                dd.big_scale = dd.big_shift = dd.small_scale = dd.small_shift = None
                setattr(dd, side + "_scale", sc.data_ptr()); setattr(dd, side + "_shift", sh.data_ptr())
                setattr(dd, side + "_affine_channels", ch)
            check(lib.veex_wgrad_gemm(C.byref(d), _lib.current_stream()), label)
            check(lib.veex_wgrad_gemm(C.byref(d3), _lib.current_stream()), label)
            torch.cuda.synchronize()
            g, g3 = ref.view(d.nsplit, M, d.Cs).sum(0), out.view(d3.nsplit, M, d3.Cs).sum(0)
            assert (g3 - g).abs().max().item() <= 5e-6 * g.abs().max().item(), (label, side)
    assert n == 6

