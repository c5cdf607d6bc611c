"""Sweep the plan of the halo-tile bf16 kernel (halo_gemm_bf16.hip) per launch of one train step: samples per
workgroup / bands per sample, resident channel slice, channel tile, register bound -- each forced through the
VEEX_HG_PLAN switch, timed with HIP events beside the general bf16 kernel (VEEX_HALO=0) in ONE process.
usage: python tools/hg_sweep.py [batch] [label-substring ...]"""
import ctypes as C
import itertools
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from vae_eeg_xai_amd import _lib, roofline as rl  # noqa: E402

lib = _lib.lib
lib.veexi_debug_hg_plan.restype = C.c_int
lib.veexi_debug_hg_plan.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]


def timed(fn, iters=12):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / iters)
    return best


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    want = sys.argv[2:]
    for label, macs, fn in rl.gemm_cases(B, True):
        if label.endswith("wgrad") or (want and not any(w in label for w in want)):
            continue
        d = fn.__defaults__[0]
        _lib.setenv("VEEX_HG_PLAN", None)
        _lib.setenv("VEEX_HALO", "0")
        t_gen = timed(fn)
        _lib.setenv("VEEX_HALO", None)
        out = (C.c_longlong * 8)()
        res = []
        if lib.veexi_debug_hg_plan(C.byref(d), out) == 1:
            res.append((timed(fn), "model's choice", tuple(out)))
        seen = set()
        for G, nb, CH, NT in itertools.product((1, 2, 3, 4, 6, 8, 12, 16), (1, 2, 3, 4, 6), (64, 128, 256), (1, 2)):
            if G > 1 and nb > 1:
                continue
            _lib.setenv("VEEX_HG_PLAN", _lib.hg_plan_code(G, nb, CH, NT))
            if lib.veexi_debug_hg_plan(C.byref(d), out) != 1:
                continue
            plan = tuple(out)[:7]
            if plan in seen:
                continue
            seen.add(plan)
            for occ in ((4, 2) if plan[3] * plan[4] <= 3 else (2,)):
                _lib.setenv("VEEX_HG_PLAN", _lib.hg_plan_code(G, nb, CH, NT, occ))
                res.append((timed(fn), f"G={G} bands={nb} CH={CH} NT={NT} occ={occ}", tuple(out)))
        _lib.setenv("VEEX_HG_PLAN", None)
        res.sort(key=lambda r: r[0])
        print(f"{label:20s} general {t_gen:7.1f} us | " + (" | ".join(f"{t:6.1f} us {c} (mt {p[3]} lds {p[5] // 1024}K wgs {p[6]})" for t, c, p in res[:4]) if res else "halo kernel does not take it"), flush=True)
        for t, c, p in res:
            if c == "model's choice":
                print(f"{'':20s} model's choice: {t:6.1f} us  G={p[0]} bands={p[1]} CH={p[2]} mt={p[3]} nt={p[4]} lds={p[5]} wgs={p[6]} cost={p[7]}", flush=True)


if __name__ == "__main__":
    main()
