"""GPU: the lean fp32 gather kernel against the general one, launch by launch through the operator-level C-ABI
(veex_gather_gemm), at the full training batch -- where its tiles run over (image row, sample, column), taps that
are padding for a whole tile are skipped and the blocks are dealt per sample group over the XCDs.  Skipped products
are exact zeros and the k order is the same, so the outputs must be BIT-identical; the per-tile column partials sit
in other tiles, so their column totals are compared (fp32 rounding of a different summation order)."""
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
    saved = {v: os.environ.get(v) for v in ("VEEX_GGV3", "VEEX_ROWORD")}
    res = []
    try:
        for v3, ro in (("0", "0"), ("1", "0"), ("1", "1")):
            os.environ["VEEX_GGV3"], os.environ["VEEX_ROWORD"] = v3, ro
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
            if val is None:
                os.environ.pop(v, None)
            else:
                os.environ[v] = val
    ref = res[0][0]
    assert torch.isfinite(ref).all() and float(ref.abs().max()) > 0
    assert torch.equal(ref, res[1][0]), "lean kernel, sample-major tiles"
    assert torch.equal(ref, res[2][0]), "lean kernel, row-major tiles with skipped padding taps"
    if epi == 1:
        assert torch.equal(res[0][1], res[1][1])
        ta, tc = res[0][1].view(mt, 3, Cout).sum(0), res[2][1].view(mt, 3, Cout).sum(0)
        assert float((ta - tc).abs().max()) <= 1e-5 * float(ta.abs().max())
