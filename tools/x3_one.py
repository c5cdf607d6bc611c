"""One layer of the split-fp32 gather kernel, a few launches (for rocprofv3 --pmc): python tools/x3_one.py label cfg nst"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vae_eeg_xai_amd import _lib, roofline as rl
from vae_eeg_xai_amd._lib import lib, check, GemmDesc
want, cfg, nst = sys.argv[1], sys.argv[2], sys.argv[3]
torch.zeros(1, device="cuda")
for label, macs, fn in rl.gemm_cases(512, False):
    if want not in label:
        continue
    d = fn.__defaults__[0]
    W = next(t for t in rl.gemm_cases._keep if isinstance(t, torch.Tensor) and t.data_ptr() == d.weight)
    w3 = torch.empty(3 * W.numel(), dtype=torch.bfloat16, device="cuda")
    check(lib.veex_pack_weights_bf16x3(_lib.ptr(W), _lib.ptr(w3), d.kh * d.kw, W.shape[2], W.shape[3], 1 if d.mode == 0 else 0, _lib.current_stream()), "pack3")
    d2 = GemmDesc.from_buffer_copy(d); d2.weight_bf16 = w3.data_ptr(); d2.weight_planes = 3; d2.weight_t = None
    _lib.setenv("VEEX_X3_CFG", cfg); _lib.setenv("VEEX_X3_NST", nst)
    for _ in range(4):
        check(lib.veex_gather_gemm(C.byref(d2), _lib.current_stream()), label)
    torch.cuda.synchronize()
