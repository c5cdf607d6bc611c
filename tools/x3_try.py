"""Split-fp32 (three bf16 terms, six partial products) gather kernel against the fp32-MFMA kernel: per launch time and
the difference of the results, per tile configuration (VEEX_X3_CFG: a `make TIMING=1` build; the shipped library runs its
table whatever the variable says).  usage: python tools/x3_try.py [batch]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
if os.environ.get('VEEX_LIB'):
    import shutil; shutil.copy(os.environ['VEEX_LIB'], os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'vae_eeg_xai_amd', 'libveex.so'))
from vae_eeg_xai_amd import _lib, roofline as rl  # noqa: E402
from vae_eeg_xai_amd._lib import lib, check, GemmDesc  # noqa: E402


def timed(fn, reps=20, groups=3):
    ts = []
    for _ in range(groups):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn(); torch.cuda.synchronize()
        a.record()
        for _ in range(reps):
            fn()
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / reps * 1e3)
    return sorted(ts)[len(ts) // 2]


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    torch.zeros(1, device="cuda")
    tot32 = tot3 = 0.0
    for label, macs, fn in rl.gemm_cases(B, False):
        if label.endswith("wgrad") or (os.environ.get('X3_ONLY') and os.environ['X3_ONLY'] not in label):
            continue
        d = fn.__defaults__[0]
        shape = (d.N, d.Hout, d.Wout, d.Cout)
        out = torch.empty(shape, device="cuda"); d.out = out.data_ptr()
        t32 = timed(fn)
        ref = out.clone()
        k, rows_cols = d.kh, None
        # weights (k,k,rows,cols): Conv2D (k,k,Cin,Cout), Conv2DTranspose (k,k,Cout,Cin); GEMM wants [tap][Cout_gemm][Cin_gemm]
        W = next(t for t in rl.gemm_cases._keep if isinstance(t, torch.Tensor) and t.data_ptr() == d.weight)
        rows, cols = W.shape[2], W.shape[3]
        transpose = 1 if d.mode == 0 else 0
        w3 = torch.empty(3 * W.numel(), dtype=torch.bfloat16, device="cuda")
        check(lib.veex_pack_weights_bf16x3(_lib.ptr(W), _lib.ptr(w3), k * k, rows, cols, transpose, _lib.current_stream()), "pack3")
        d2 = GemmDesc.from_buffer_copy(d)
        d2.weight_bf16 = w3.data_ptr(); d2.weight_planes = 3; d2.weight_t = None
        out2 = torch.full(shape, 7.0, device="cuda"); d2.out = out2.data_ptr()
        f2 = lambda: check(lib.veex_gather_gemm(C.byref(d2), _lib.current_stream()), label)
        res = []
        opts = [int(o) for o in os.environ.get('X3_OPTS', '0').split(',')]      # VEEX_X3_OPT: 1 swizzled rows, 2 store first
        for cfg, opt in [(int(c), o) for c in os.environ.get('X3_CFGS', '-2,0,2,3,4,5,6').split(',') for o in opts]:   # -2 = the shipped table
            _lib.setenv("VEEX_X3_CFG", cfg)
            _lib.setenv("VEEX_X3_OPT", opt if opt else None)
            if d.epilogue == 1:
                mt = lib.veex_gather_gemm_m_tiles(C.byref(d2))
            out2.fill_(7.0)
            t3 = timed(f2)
            torch.cuda.synchronize()
            diff = (out2 - ref).abs().max().item() / ref.abs().max().item()
            res.append((t3, f"cfg{cfg}" + (f"/opt{opt}" if len(opts) > 1 else ""), diff))
        _lib.setenv("VEEX_X3_CFG", None)
        _lib.setenv("VEEX_X3_OPT", None)
        t3 = min(res)[0]
        tot32 += t32; tot3 += t3
        print(f"{label:20s} fp32 {t32:7.1f} us | " + " | ".join(f"{c} {t:6.1f}" + ("" if df < 5e-6 else f" BAD {df:.1e}") for t, c, df in res), flush=True)
    print(f"sum fp32 {tot32:.1f} us   x3 {tot3:.1f} us")


if __name__ == "__main__":
    main()
