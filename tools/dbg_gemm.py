import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vae_eeg_xai_amd import _lib
from vae_eeg_xai_amd._lib import lib
from vae_eeg_xai_amd.roofline import _time_launch, _rand

class GGParams(C.Structure):
    _fields_ = [("A", C.c_void_p), ("W", C.c_void_p), ("Wbf", C.c_void_p), ("Out", C.c_void_p), ("a_scale", C.c_void_p), ("a_shift", C.c_void_p),
                ("bias", C.c_void_p), ("aux", C.c_void_p), ("e_scale", C.c_void_p), ("partials", C.c_void_p),
                ("Nb", C.c_int), ("Hin", C.c_int), ("Win", C.c_int), ("Cin", C.c_int), ("Hout", C.c_int), ("Wout", C.c_int), ("Cout", C.c_int),
                ("kh", C.c_int), ("kw", C.c_int), ("stride", C.c_int), ("pb", C.c_int), ("mode", C.c_int),
                ("aff_mask", C.c_int), ("aff_spatial", C.c_int), ("epi", C.c_int), ("relu", C.c_int), ("epi_cmask", C.c_int),
                ("nsplit", C.c_int), ("chunks_per_split", C.c_int), ("nphase", C.c_int), ("tile_prefix", C.c_int * 5),
                ("Hp", C.c_int * 2), ("Wp", C.c_int * 2), ("o0", C.c_int * 2), ("slab_stride", C.c_longlong), ("gx", C.c_int), ("gy", C.c_int), ("gz", C.c_int), ("remap", C.c_int), ("tap_inner", C.c_int),
                ("Abf", C.c_void_p), ("OutBf", C.c_void_p), ("o_scale", C.c_void_p), ("o_shift", C.c_void_p), ("o_cmask", C.c_int), ("dbg", C.c_int)]

raw = C.CDLL(_lib.LIB_PATH)
fn = raw.veexi_debug_gather_gemm
fn.argtypes = [C.POINTER(GGParams), C.c_int, C.c_void_p]; fn.restype = C.c_int
B = 512
def case(mode, hin, cin, hout, cout, k, s, pb, aff):
    x = _rand(B, hin, hin, cin); y = torch.empty(B, hout, hout, cout, device="cuda")
    W = _rand(k, k, cin, cout) * 0.1; bias = _rand(cout)
    sc, sh = torch.rand(cin, device="cuda") + 0.5, _rand(cin)
    p = GGParams()
    p.A, p.W, p.Out, p.bias = x.data_ptr(), W.data_ptr(), y.data_ptr(), bias.data_ptr()
    if aff: p.a_scale, p.a_shift, p.aff_mask = sc.data_ptr(), sh.data_ptr(), cin - 1
    p.Nb, p.Hin, p.Win, p.Cin, p.Hout, p.Wout, p.Cout = B, hin, hin, cin, hout, hout, cout
    p.kh = p.kw = k; p.stride, p.pb, p.mode, p.epi, p.relu = s, pb, mode, 0, 1
    keep = (x, y, W, bias, sc, sh)
    return p, keep
for label, args in [("deconv4.fwd   (mode1 64x64 aff)  ", (1, 17, 128, 19, 64, 3, 1, 0, True)),
                    ("deconv4.dgrad (mode0 64x64)      ", (0, 19, 64, 17, 128, 3, 1, 0, False)),
                    ("deconv2.dgrad (mode0 64x64 C=128)", (0, 8, 128, 8, 256, 3, 1, 1, False))]:
    p, keep = case(*args)
    for dbg, nm in [(0, "full"), (16, "no-epilogue"), (64, "A loads hit 4KiB"), (1, "no-gload"), (17, "no-gload, no-epilogue"), (13, "mfma+ldsread only"), (29, "mfma+ldsread, no-epilogue")]:
        t = _time_launch(lambda: fn(C.byref(p), dbg, _lib.current_stream()), iters=10)
        print(f"{label} {nm:26s} {t*1e6:8.1f} us")
pk = raw.veexi_debug_mfma_peak; pk.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]; pk.restype = C.c_int
out = torch.empty(4096 * 256, device="cuda")
for blocks in (256, 512, 768):
    iters = 2000
    t = _time_launch(lambda: pk(out.data_ptr(), blocks, iters, _lib.current_stream()), iters=5)
    fl = blocks * 4 * iters * 32 * 4096.0
    print(f"pure MFMA {blocks} blocks: {t*1e6:.1f} us  {fl/t/1e12:.1f} TF")
