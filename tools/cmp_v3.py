"""GPU: the lean 64x64 fp32 gather kernel (VEEX_GGV3=1) against the general one (VEEX_GGV3=0), launch by launch,
through the operator-level C-ABI.  Prints max |diff| / max |ref| per case (both run the same k order: 0 expected)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vae_eeg_xai_amd import _lib
from vae_eeg_xai_amd._lib import lib, check, GemmDesc

torch.manual_seed(0)
def R(*s): return torch.rand(*s, device="cuda") - 0.5

def run(name, mode, N, Hin, Cin, Hout, Cout, k, s, pb, aff=0, epi=0, nsplit=1):
    x = R(N, Hin, Hin, Cin)
    W = R(k, k, Cin, Cout) if mode == 0 else R(k, k, Cout, Cin)
    bias = R(Cout)
    outs = []
    d = GemmDesc()
    d.inp, d.weight, d.bias = x.data_ptr(), W.data_ptr(), bias.data_ptr()
    keep = []
    if aff == 1:
        sc, sh = torch.rand(Cin, device="cuda") + 0.5, R(Cin); keep += [sc, sh]
        d.in_scale, d.in_shift, d.affine_channels, d.affine_spatial = sc.data_ptr(), sh.data_ptr(), Cin, 0
    elif aff == 2:
        n = Hin * Hin * Cin
        sc, sh = torch.rand(n, device="cuda") + 0.5, R(n); keep += [sc, sh]
        d.in_scale, d.in_shift, d.affine_channels, d.affine_spatial = sc.data_ptr(), sh.data_ptr(), n, 1
    d.N, d.Hin, d.Win, d.Cin, d.Hout, d.Wout, d.Cout = N, Hin, Hin, Cin, Hout, Hout, Cout
    d.kh = d.kw = k; d.stride, d.pad_begin, d.mode = s, pb, mode
    d.epilogue, d.relu, d.nsplit = epi, 1, nsplit
    if epi == 1:
        aux = R(N, Hout, Hout, Cout); esc = torch.rand(Cout, device="cuda") + 0.5; keep += [aux, esc]
        d.aux, d.e_scale, d.e_channels = aux.data_ptr(), esc.data_ptr(), Cout
        tmp = torch.empty(1, device="cuda"); d.out = tmp.data_ptr()
        mt = lib.veex_gather_gemm_m_tiles(C.byref(d))
    res = []
    for v, ro in (("0", "0"), ("1", "0"), ("1", "1")):
        _lib.setenv("VEEX_GGV3", v)
        _lib.setenv("VEEX_ROWORD", ro)   # 1: tiles over (image row, sample, column); skips taps a whole tile pads
        out = torch.full((max(nsplit, 1), N, Hout, Hout, Cout), 7.0, device="cuda")
        d.out = out.data_ptr()
        part = None
        if epi == 1:
            part = torch.zeros(mt * 3 * Cout, device="cuda"); d.partials = part.data_ptr()
        check(lib.veex_gather_gemm(C.byref(d), _lib.current_stream()), name)
        torch.cuda.synchronize()
        res.append((out.clone(), None if part is None else part.clone()))
    a, b, c = res[0][0], res[1][0], res[2][0]
    den = max(a.abs().max().item(), 1e-30)
    msg = f"{name:34s} out rel diff {(a - b).abs().max().item() / den:.3e} / row order 1: {(a - c).abs().max().item() / den:.3e}"
    if epi == 1:
        pa, pb_, pc = res[0][1], res[1][1], res[2][1]
        pden = max(pa.abs().max().item(), 1e-30)
        ta, tc = pa.view(mt, 3, Cout).sum(0), pc.view(mt, 3, Cout).sum(0)   # tiles hold other rows: compare the totals
        msg += (f"   partials rel diff {(pa - pb_).abs().max().item() / pden:.3e}"
                f" / column totals, row order 1: {(ta - tc).abs().max().item() / max(ta.abs().max().item(), 1e-30):.3e}")
    print(msg)

run("5x5 valid Cout=128 fwd", 0, 4, 5, 256, 1, 128, 5, 1, 0, aff=1)
run("5x5 valid Cout=100 fwd", 0, 4, 5, 256, 1, 100, 5, 1, 0, aff=1)
run("5x5 valid Cout=100 fwd noaff", 0, 4, 5, 256, 1, 100, 5, 1, 0, aff=0)
run("3x3 valid Cout=100 fwd", 0, 4, 5, 256, 3, 100, 3, 1, 0, aff=1)
run("4x4 valid Cout=128 fwd", 0, 4, 5, 256, 2, 128, 4, 1, 0, aff=1)
run("5x5 valid Cin=64 Cout=128 fwd", 0, 4, 5, 64, 1, 128, 5, 1, 0, aff=1)
for N in (4, 64):
    run(f"conv2 fwd N={N}", 0, N, 20, 64, 10, 128, 3, 2, 0, aff=1)
    run(f"conv3 fwd N={N}", 0, N, 10, 128, 5, 256, 3, 2, 0, aff=1)
    run(f"enc.dense fwd split N={N}", 0, N, 5, 256, 1, 100, 5, 1, 0, aff=1, epi=2, nsplit=8)
    run(f"enc.dense fwd nosplit N={N}", 0, N, 5, 256, 1, 100, 5, 1, 0, aff=1, epi=2, nsplit=1)
    run(f"deconv1 fwd N={N}", 1, N, 4, 256, 8, 256, 3, 2, 0, aff=2)
    run(f"deconv2 fwd N={N}", 1, N, 8, 256, 8, 128, 3, 1, 1, aff=1)
    run(f"deconv3 fwd N={N}", 1, N, 8, 128, 17, 128, 3, 2, 0, aff=1)
    run(f"deconv4 fwd N={N}", 1, N, 17, 128, 19, 64, 3, 1, 0, aff=1)
    run(f"deconv5 fwd N={N}", 1, N, 19, 64, 39, 64, 3, 2, 0, aff=1)
    run(f"deconv5 dgrad N={N}", 0, N, 39, 64, 19, 64, 3, 2, 0, epi=1)
    run(f"deconv4 dgrad N={N}", 0, N, 19, 64, 17, 128, 3, 1, 0, epi=1)
    run(f"deconv2 dgrad N={N}", 0, N, 8, 128, 8, 256, 3, 1, 1, epi=1)
    run(f"deconv1 dgrad N={N}", 0, N, 8, 256, 4, 256, 3, 2, 0, epi=0)
    run(f"conv3 dgrad N={N}", 1, N, 5, 256, 10, 128, 3, 2, 0, epi=1)
    run(f"conv2 dgrad N={N}", 1, N, 10, 128, 20, 64, 3, 2, 0, epi=1)
    run(f"dz split N={N}", 1, N, 1, 4096, 1, 28, 1, 1, 0, epi=2, nsplit=8)
