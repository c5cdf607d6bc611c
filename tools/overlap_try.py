"""Does running a layer's weight-gradient GEMM on a second stream next to its input-gradient GEMM pay?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vae_eeg_xai_amd import roofline as rl, _lib

cases = {l: fn for l, _, fn in rl.gemm_cases(512)}
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def run_on(stream, fn):
    with torch.cuda.stream(stream):
        fn()
def timeit(f, iters=20):
    for _ in range(5): f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e6
for layer in ("dec.deconv5", "dec.deconv4", "dec.deconv3", "dec.deconv2", "enc.conv3", "enc.conv2"):
    d, w = cases[layer + ".dgrad"], cases[layer + ".wgrad"]
    seq = timeit(lambda: (run_on(s1, d), run_on(s1, w)))
    def both():
        run_on(s1, d); run_on(s2, w)
    par = timeit(both)
    print(f"{layer:12s} sequential {seq:7.1f} us   two streams {par:7.1f} us   ({100*(par/seq-1):+.1f} %)")
