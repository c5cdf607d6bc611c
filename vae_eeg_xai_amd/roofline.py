"""Per-kernel timing of the hot path's GEMM launches through the operator-level C-ABI, with HIP
events on the launch stream, and the roofline arithmetic bench.py reports (Measurement).

Algorithmic work of one launch = exact in-bounds MACs of SURVEY 8a (x batch) for the layer; the
bound for every conv / deconv / dense contraction in fp32 is the f32 MFMA peak (157.3 TFLOP/s,
MI355X_MICROARCH.md); the HBM-bound kernels are priced in bytes against 8 TB/s.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import _lib
from ._lib import lib, check, GemmDesc, WgradDesc

PEAK_F32_MFMA_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0
PEAK_BF16_MFMA_TFLOPS = 2500.0

# (name, Cin, Cout, k, stride, pad_begin, Hin, Hout, exact forward MACs per sample [SURVEY 8a])
CONVS = [
    ("enc.conv2", 64, 128, 3, 2, 0, 20, 10, 6_889_472),
    ("enc.conv3", 128, 256, 3, 2, 0, 10, 5, 6_422_528),
]
DECONVS = [
    ("dec.deconv1", 256, 256, 3, 2, 0, 4, 8, 7_929_856),
    ("dec.deconv2", 256, 128, 3, 1, 1, 8, 8, 15_859_712),
    ("dec.deconv3", 128, 128, 3, 2, 0, 8, 17, 9_437_184),
    ("dec.deconv4", 128, 64, 3, 1, 0, 17, 19, 21_307_392),
    ("dec.deconv5", 64, 64, 3, 2, 0, 19, 39, 13_307_904),
]


def _time_launch(fn, iters=10, warm=2, rounds=3, warm_ms=30.0):
    """Seconds per launch: HIP events on the launch stream; the chip is first kept busy for warm_ms
    (an idle MI355X drops its clock and needs milliseconds of work to raise it again), then the
    median of `rounds` timed groups of `iters` back-to-back launches is taken."""
    import time
    st = torch.cuda.current_stream()
    for _ in range(warm):
        fn()
    st.synchronize()
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < warm_ms:
        for _ in range(4):
            fn()
        st.synchronize()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(iters):
            fn()
        e1.record(st)
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e-3 / iters)
    ts.sort()
    return ts[len(ts) // 2]     # seconds per launch


def _rand(*shape):
    return torch.rand(*shape, device="cuda", dtype=torch.float32) - 0.5


def _pack(W, transpose):
    """bf16 copy [tap][Cout_gemm][Cin_gemm] of a (k,k,rows,cols) kernel (veex_pack_weights_bf16)."""
    k, _, rows, cols = W.shape
    dst = torch.empty(W.numel(), dtype=torch.bfloat16, device="cuda")
    check(lib.veex_pack_weights_bf16(_lib.ptr(W), _lib.ptr(dst), k * k, rows, cols, transpose, _lib.current_stream()), "pack")
    return dst


def _transposed(W):
    """fp32 copy [tap][cols][rows] of a (k,k,rows,cols) kernel (veex_transpose_weights_f32), as the whole-step entry
    points make for the mode-0 launches of the lean fp32 kernel."""
    k, _, rows, cols = W.shape
    dst = torch.empty(W.numel(), dtype=torch.float32, device="cuda")
    check(lib.veex_transpose_weights_f32(_lib.ptr(W), _lib.ptr(dst), k * k, rows, cols, _lib.current_stream()), "transpose")
    return dst


def _pack3(W, transpose):
    """three-term bf16 split [tap][Cin/16][plane][Cout][16] of a (k,k,rows,cols) kernel (veex_pack_weights_bf16x3)."""
    k, _, rows, cols = W.shape
    dst = torch.empty(3 * W.numel(), dtype=torch.bfloat16, device="cuda")
    check(lib.veex_pack_weights_bf16x3(_lib.ptr(W), _lib.ptr(dst), k * k, rows, cols, transpose, _lib.current_stream()), "pack3")
    return dst


def gemm_cases(B, bf16=False, x3=False):
    """Yield (label, macs, launch_fn) for every MFMA launch of one train step at batch B.  x3: the split-fp32 engine
    (forward and input-gradient launches carry the three-plane weights, weight gradients ask for the split kernel)."""
    stream = _lib.current_stream
    keep = []
    # fp32 engine: producers write a BatchNorm-applied fp32 copy (VeexGemmDesc::out_bn) and consumers read it without an
    # affine prologue, as the whole-step entry points do
    bncopy = not bf16

    def gg(label, macs, mode, A, W, out, N, Hin, Cin, Hout, Cout, k, stride, pb, aff=None, spatial=0, epi=0,
           aux=None, esc=None, wbf=None, osc=None):
        d = GemmDesc()
        d.inp, d.weight, d.out = A.data_ptr(), W.data_ptr(), out.data_ptr()
        if wbf is None and mode == 0:
            wt = _transposed(W); keep.append(wt); d.weight_t = wt.data_ptr()
        if x3:
            w3 = _pack3(W, 1 if mode == 0 else 0); keep.append(w3)
            d.weight_bf16, d.weight_planes = w3.data_ptr(), 3
        if wbf is not None:
            keep.append(wbf); d.weight_bf16 = wbf.data_ptr()
            # bf16 consumer copies as the network uses them: A with its BatchNorm applied, Out copy for the next layer
            if aff is not None and not spatial:
                abf = (A * aff[0] + aff[1]).to(torch.bfloat16)
            elif aff is not None:
                abf = (A.reshape(N, -1) * aff[0] + aff[1]).reshape(A.shape).to(torch.bfloat16)
            else:
                abf = A.to(torch.bfloat16)
            obf = torch.empty(out.shape, dtype=torch.bfloat16, device="cuda")
            keep.extend([abf, obf]); d.in_bf16, d.out_bf16 = abf.data_ptr(), obf.data_ptr()
            if osc is not None:
                keep.append(osc); d.out_scale, d.out_shift, d.out_channels = osc[0].data_ptr(), osc[1].data_ptr(), osc[0].numel()
        if aff is not None and not (bncopy and epi == 0):
            d.in_scale, d.in_shift, d.affine_channels, d.affine_spatial = aff[0].data_ptr(), aff[1].data_ptr(), aff[0].numel(), spatial
        if bncopy and epi == 0 and osc is not None and label != "dec.deconv5.fwd":   # deconv6 applies BatchNorm 5 itself
            obn = torch.empty_like(out)
            keep.extend([obn, osc]); d.out_bn = obn.data_ptr()
            d.out_scale, d.out_shift, d.out_channels = osc[0].data_ptr(), osc[1].data_ptr(), osc[0].numel()
        if epi == 0:
            bias = _rand(Cout); keep.append(bias); d.bias = bias.data_ptr(); d.relu = 1
        else:
            d.aux, d.relu = aux.data_ptr(), 1
            if esc is not None:
                d.e_scale, d.e_channels = esc.data_ptr(), esc.numel()
            mt = None
        d.N, d.Hin, d.Win, d.Cin, d.Hout, d.Wout, d.Cout = N, Hin, Hin, Cin, Hout, Hout, Cout
        d.kh = d.kw = k
        d.stride, d.pad_begin, d.mode, d.epilogue, d.nsplit = stride, pb, mode, epi, 1
        if epi == 1:
            # sized for the smallest tile (64 rows, 4 phases) so that tile-size switches flipped by
            # environment variables between launches can never overrun it
            mt = (N * Hout * Hout + 31) // 32 + 4
            assert lib.veex_gather_gemm_m_tiles(C.byref(d)) <= mt
            part = torch.empty(mt * 3 * Cout, device="cuda"); keep.append(part); d.partials = part.data_ptr()
        keep.extend([A, W, out, d, aff, esc, aux])   # (everything a descriptor points to must outlive gemm_cases)
        return label, macs, (lambda d=d: check(lib.veex_gather_gemm(C.byref(d), stream()), label))

    def wg(label, macs, big, small, N, Hb, Cb, Hs, Cs, k, stride, pb, aff_small=None, aff_big=None):
        d = WgradDesc()
        d.big, d.small = big.data_ptr(), small.data_ptr()
        if bncopy:
            aff_small = aff_big = None
        if aff_small is not None:
            d.small_scale, d.small_shift, d.small_affine_channels = aff_small[0].data_ptr(), aff_small[1].data_ptr(), aff_small[0].numel()
        if aff_big is not None:
            d.big_scale, d.big_shift, d.big_affine_channels = aff_big[0].data_ptr(), aff_big[1].data_ptr(), aff_big[0].numel()
        d.N, d.Hb, d.Wb, d.Cb, d.Hs, d.Ws, d.Cs = N, Hb, Hb, Cb, Hs, Hs, Cs
        d.kh = d.kw = k
        d.stride, d.pad_begin = stride, pb
        M = k * k * Cb
        d.bf16 = 1 if bf16 else (2 if x3 else 0)     # 2: the split-fp32 weight-gradient kernel
        if bf16:
            bb = (big * aff_big[0] + aff_big[1]) if aff_big is not None else big
            sb = (small * aff_small[0] + aff_small[1]) if aff_small is not None else small
            bb, sb = bb.to(torch.bfloat16), sb.to(torch.bfloat16)
            keep.extend([bb, sb]); d.big_bf16, d.small_bf16 = bb.data_ptr(), sb.data_ptr()
        ns, per = C.c_int32(), C.c_int32()   # the row split the whole-step entry point uses
        check(lib.veex_wgrad_plan(C.byref(d), C.byref(ns), C.byref(per)), "wgrad_plan")
        d.nsplit, d.rows_per_split = ns.value, per.value
        out = torch.empty(d.nsplit * M * Cs, device="cuda")
        d.out = out.data_ptr()
        keep.extend([big, small, out, d, aff_small, aff_big])
        return label, macs, (lambda d=d: check(lib.veex_wgrad_gemm(C.byref(d), stream()), label))

    cases = []
    for name, cin, cout, k, s, pb, hin, hout, macs in CONVS:
        x = _rand(B, hin, hin, cin); y = torch.empty(B, hout, hout, cout, device="cuda")
        W = _rand(k, k, cin, cout) * 0.1
        aff = (torch.rand(cin, device="cuda") + 0.5, _rand(cin))
        dy = _rand(B, hout, hout, cout); dx = torch.empty_like(x)
        wt, wa = (_pack(W, 1), _pack(W, 0)) if bf16 else (None, None)
        oaff = (torch.rand(cout, device="cuda") + 0.5, _rand(cout))
        cases.append(gg(f"{name}.fwd", macs * B, 0, x, W, y, B, hin, cin, hout, cout, k, s, pb, aff=aff, wbf=wt, osc=oaff))
        cases.append(gg(f"{name}.dgrad", macs * B, 1, dy, W, dx, B, hout, cout, hin, cin, k, s, pb, epi=1, aux=x,
                        esc=aff[0], wbf=wa))
        cases.append(wg(f"{name}.wgrad", macs * B, x, dy, B, hin, cin, hout, cout, k, s, pb, aff_big=aff))
    for name, cin, cout, k, s, pb, hin, hout, macs in DECONVS:
        x = _rand(B, hin, hin, cin); y = torch.empty(B, hout, hout, cout, device="cuda")
        W = _rand(k, k, cout, cin) * 0.1
        spatial = 1 if name == "dec.deconv1" else 0
        nch = hin * hin * cin if spatial else cin
        aff = (torch.rand(nch, device="cuda") + 0.5, _rand(nch))
        dy = _rand(B, hout, hout, cout); dx = torch.empty_like(x)
        wt, wa = (_pack(W, 1), _pack(W, 0)) if bf16 else (None, None)
        oaff = (torch.rand(cout, device="cuda") + 0.5, _rand(cout))
        cases.append(gg(f"{name}.fwd", macs * B, 1, x, W, y, B, hin, cin, hout, cout, k, s, pb, aff=aff, spatial=spatial, wbf=wa, osc=oaff))
        if spatial:
            cases.append(gg(f"{name}.dgrad", macs * B, 0, dy, W, dx, B, hout, cout, hin, cin, k, s, pb, wbf=wt))
        else:
            cases.append(gg(f"{name}.dgrad", macs * B, 0, dy, W, dx, B, hout, cout, hin, cin, k, s, pb, epi=1, aux=x,
                            esc=aff[0], wbf=wt))
        if not spatial or bf16:
            cases.append(wg(f"{name}.wgrad", macs * B, dy, x, B, hout, cout, hin, cin, k, s, pb,
                            aff_small=None if spatial else aff))
    gemm_cases._keep = keep
    return cases


def algorithmic_bytes(label, B, bf16=False):
    """HBM bytes one launch of gemm_cases() must move: every operand read once, every result written once, in the
    element types the engine uses (fp32 engine: everything fp32; bf16 engine: operands are bf16 consumer copies,
    results are written as fp32 AND as the bf16 copy the next GEMM reads; a backward epilogue also reads the fp32
    activation; a weight gradient writes the fp32 gradient once -- split-K slabs are overhead, not algorithm)."""
    layer, kind = label.rsplit(".", 1)
    for name, cin, cout, k, s, pb, hin, hout, macs in CONVS + DECONVS:
        if name == layer:
            break
    else:
        raise KeyError(label)
    n_in, n_out, n_w = B * hin * hin * cin, B * hout * hout * cout, k * k * cin * cout
    e = 2 if bf16 else 4                      # operand element size
    ocopy = 2 if bf16 else 0                  # extra bf16 copy of a result
    if kind == "fwd":
        if not bf16 and layer != "dec.deconv5":
            ocopy = 4                         # the BatchNorm-applied fp32 copy the next GEMMs read
        return n_in * e + n_w * e + n_out * (4 + ocopy)
    if kind == "dgrad":                       # reads d(out) and the activation of the layer below, writes d(in)
        aux = 0 if (layer == "dec.deconv1") else n_in * 4
        return n_out * e + n_w * e + aux + n_in * (4 + ocopy)
    return n_in * e + n_out * e + n_w * 4     # wgrad


PEAK_X3_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 6.0     # split-fp32: six bf16 MFMAs per fp32 multiply-accumulate


def layer_table(B=512, iters=10, bf16=False, x3=False):
    """x3: the split-fp32 engine; its forward / input-gradient launches are priced against the bf16 matrix pipe at six
    MFMAs per product (416.7 TFLOP/s of fp32 work), its weight gradients against the fp32 MFMA peak they run on."""
    rows = []
    for label, macs, fn in gemm_cases(B, bf16, x3):
        peak = PEAK_BF16_MFMA_TFLOPS if bf16 else (PEAK_X3_TFLOPS if x3 else PEAK_F32_MFMA_TFLOPS)
        t = _time_launch(fn, iters)
        tf = 2.0 * macs / t / 1e12
        nbytes = algorithmic_bytes(label, B, bf16)
        t_roof = max(2.0 * macs / (peak * 1e12), nbytes / (PEAK_HBM_GBS * 1e9))
        rows.append({"kernel": label, "us": t * 1e6, "tflops": tf, "frac": tf / peak, "bytes": nbytes,
                     "roof_us": t_roof * 1e6, "roof": "mfma" if 2.0 * macs / (peak * 1e12) >= nbytes / (PEAK_HBM_GBS * 1e9) else "hbm",
                     "frac_roof": t_roof / t})
    return rows


_PMC_NOTE = ("HBM bytes per launch (read + write) from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes "
             "(tools/run_profiles.sh, batch 512; FETCH_SIZE doubled per the gfx950 correction), profiles/*_pmc_gemms.json")


def _pmc_traffic(label, bf16=False, x3=False):
    """Per-launch HBM traffic of a GEMM from the newest committed PMC summary (counters cannot be
    collected inside bench.py: they need rocprofv3 passes of their own)."""
    import glob, json, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = sorted(glob.glob(os.path.join(root, "profiles", "r*_pmc_gemms_bf16.json" if bf16 else "r*_pmc_gemms_x3.json" if x3 else "r*_pmc_gemms.json")))
    if not files:
        return None
    try:
        e = json.load(open(files[-1]))[label]
        return e["hbm_read_bytes"] + e["hbm_write_bytes"]
    except (KeyError, ValueError, OSError):
        return None


def _time_in_step(net, batch, layer, steps=20, warm=3):
    """Average duration (s) of the forward launch of `layer` inside `steps` real train steps of `net` at `batch`
    (synthetic input, Philox epsilon, no optimizer step: the weights stay as they are)."""
    import ctypes
    lib.veexi_debug_probe_arm.argtypes = [C.c_void_p, C.c_char_p]
    lib.veexi_debug_probe_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    x = torch.rand(batch, 40, 40, 1, device="cuda")
    tot, ms = 0.0, ctypes.c_float()
    for i in range(warm + steps):
        check(lib.veexi_debug_probe_arm(net._h, layer.encode()), "probe_arm")
        net.train_fwd_bwd(x, None, accumulate_metrics=False, first_sample=0)
        check(lib.veexi_debug_probe_ms(net._h, C.byref(ms)), "probe_ms")
        if i >= warm:
            tot += ms.value
    return tot / steps * 1e-3


def dominant_kernel_roofline(net, batch):
    """The largest MFMA launch of the step (dec.deconv4 forward: the largest-FLOP layer), timed live with
    HIP events on the launch stream inside real train steps.  Split-fp32 engine: the launch runs on the bf16 matrix
    pipe at six MFMAs per fp32 multiply-accumulate, so its peak is 2500 / 6 = 416.7 TFLOP/s of fp32 work
    (``frac`` = executed bf16 MFMA FLOP / 2.5 PFLOP/s); ``frac_of_f32_mfma_peak`` prices the same rate against the
    157.3 TFLOP/s the fp32 MFMA instruction could reach at best."""
    best = None
    compute = getattr(net, "compute", "f32")
    bf16, x3 = compute == "bf16", compute == "f32x3"
    peak = PEAK_BF16_MFMA_TFLOPS if bf16 else PEAK_X3_TFLOPS if x3 else PEAK_F32_MFMA_TFLOPS
    for label, macs, fn in gemm_cases(batch, bf16, x3):
        if label != "dec.deconv4.fwd":
            continue
        t_alone = _time_launch(fn, iters=20, warm=3)
        # inside real train steps: HIP events recorded by the library around this one launch, on its launch stream
        # (veexi_debug_probe_*); this is the duration a rocprofv3 kernel trace of bench.py shows for the kernel
        t = _time_in_step(net, batch, "dec.deconv4") or t_alone
        tf = 2.0 * macs / t / 1e12
        # the committed counter passes were taken at batch 512 (tools/run_profiles.sh)
        traffic = _pmc_traffic("dec.deconv4.fwd", bf16, x3) if batch == 512 else None
        kname = "gather_gemm_bf16_kernel" if bf16 else "gather_gemm_x3_kernel<256, 64, 16> (722 workgroups of 8 waves, two per CU)" if x3 else "gather_gemm_v3_kernel"
        best = {"bound": "mfma", "kernel": kname + " mode 1 (dec.deconv4 forward, Conv2DTranspose 128->64 k3 s1)",
                "achieved": tf, "peak": peak, "unit": "TFLOP/s", "frac": tf / peak,
                "traffic": traffic, "traffic_source": _PMC_NOTE if traffic is not None else None, "us_per_launch": t * 1e6,
                "us_per_launch_standalone": t_alone * 1e6, "algorithmic_flop_per_launch": 2.0 * macs,
                "timing": "average over 20 train steps of HIP events recorded around this launch on its launch stream, inside "
                          "veex_net_train_fwd_bwd; us_per_launch_standalone = the same launch repeated back-to-back on its own"}
        if x3:
            best["peak_note"] = ("split-fp32 arithmetic: 6 v_mfma_f32_32x32x16_bf16 per 16 k of fp32 work -> 2500 / 6 TFLOP/s; "
                                 "achieved / peak = executed bf16 MFMA FLOP / 2.5 PFLOP/s")
            best["frac_of_f32_mfma_peak"] = tf / PEAK_F32_MFMA_TFLOPS
    return best


def hbm_cases(B=512, L=28):
    """(label, algorithmic bytes, launch_fn) for every HBM-bound kernel of one train step / attribution pass at
    batch B, launched stand-alone through the operator-level C-ABI.  Algorithmic bytes = each operand read once +
    each result written once; reference ops: models.py:10-22 (sample), :156-162 (BCE + KL),
    :219-223 (conv1), :300-301 (deconv6), train.py:137 (Adam)."""
    from .net import Net, ADAM_B1, ADAM_B2, ADAM_EPS
    st = _lib.current_stream
    keep = []
    F = 4

    def R(*shape):
        t = _rand(*shape); keep.append(t); return t

    def E(*shape):
        t = torch.empty(*shape, device="cuda", dtype=torch.float32); keep.append(t); return t

    cases = []
    x = torch.rand(B, 40, 40, 1, device="cuda"); keep.append(x)
    w1, b1 = R(3, 3, 1, 64), R(64)
    a1 = E(B, 20, 20, 64)
    cases.append(("conv1_fwd", (x.numel() + a1.numel()) * F,
                  lambda: check(lib.veex_conv1_fwd(x.data_ptr(), w1.data_ptr(), b1.data_ptr(), a1.data_ptr(), B, st()))))
    d1 = R(B, 20, 20, 64)
    nb = lib.veex_conv1_wgrad_blocks(B)
    slab1 = E(nb, 576)
    cases.append(("conv1_wgrad", (x.numel() + d1.numel()) * F,
                  lambda: check(lib.veex_conv1_wgrad(x.data_ptr(), d1.data_ptr(), slab1.data_ptr(), B, st()))))
    dx = E(B, 40, 40)
    cases.append(("conv1_dgrad", (d1.numel() + 2 * x.numel()) * F,
                  lambda: check(lib.veex_conv1_dgrad(d1.data_ptr(), w1.data_ptr(), x.data_ptr(), dx.data_ptr(), B, st()))))
    a5 = torch.rand(B, 39, 39, 64, device="cuda"); keep.append(a5)
    sc, sh, w6, b6 = torch.rand(64, device="cuda") + 0.5, R(64), R(2, 2, 1, 64) * 0.1, R(1)
    keep.extend([sc, sh])
    dl, rb, ds = E(B, 1600), E(B), E(B)
    cases.append(("deconv6_bce_fwd", (a5.numel() + x.numel() + dl.numel()) * F,
                  lambda: check(lib.veex_deconv6_bce_fwd(a5.data_ptr(), sc.data_ptr(), sh.data_ptr(), w6.data_ptr(),
                                                         b6.data_ptr(), x.data_ptr(), None, None, dl.data_ptr(),
                                                         rb.data_ptr(), ds.data_ptr(), B, 1.0 / B, st()))))
    d5, slab6 = E(B, 39, 39, 64), E(B, 448)
    cases.append(("deconv6_bwd", (a5.numel() + dl.numel() + d5.numel()) * F,
                  lambda: check(lib.veex_deconv6_bwd(dl.data_ptr(), a5.data_ptr(), sc.data_ptr(), sh.data_ptr(),
                                                     w6.data_ptr(), d5.data_ptr(), slab6.data_ptr(), B, st()))))
    mulv, eps, z, klb = R(B, 2 * L), R(B, L), E(B, L), E(B)
    cases.append(("reparam_kl_fwd", (mulv.numel() + 2 * eps.numel() + B) * F,
                  lambda: check(lib.veex_reparam_kl_fwd(mulv.data_ptr(), eps.data_ptr(), z.data_ptr(), klb.data_ptr(), B, L,
                                                        1, 42, 0, st()))))
    dz, dmulv = R(B, L), E(B, 2 * L)
    cases.append(("reparam_kl_bwd", (dz.numel() + mulv.numel() + eps.numel() + dmulv.numel()) * F,
                  lambda: check(lib.veex_reparam_kl_bwd(dz.data_ptr(), mulv.data_ptr(), eps.data_ptr(), dmulv.data_ptr(), B, L,
                                                        1.0 / B, st()))))
    n = 2_288_256
    p_, g_, m_, v_ = R(n), R(n), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    step, alpha = torch.zeros(1, dtype=torch.int32, device="cuda"), torch.zeros(1, device="cuda")
    keep.extend([m_, v_, step, alpha])
    cases.append(("adam_apply (+prepare)", n * 7 * F,
                  lambda: check(lib.veex_adam(p_.data_ptr(), g_.data_ptr(), m_.data_ptr(), v_.data_ptr(), n, step.data_ptr(),
                                              alpha.data_ptr(), 1e-5, 1.0, ADAM_B1, ADAM_B2, ADAM_EPS, st()))))
    # weight-gradient slab reductions as the step launches them (deconv4: 9*64 x 128 gradient)
    d = WgradDesc(); d.N, d.Hb, d.Wb, d.Cb, d.Hs, d.Ws, d.Cs = B, 19, 19, 64, 17, 17, 128
    d.kh = d.kw = 3; d.stride, d.pad_begin = 1, 0
    ns, per = C.c_int32(), C.c_int32()
    check(lib.veex_wgrad_plan(C.byref(d), C.byref(ns), C.byref(per)), "wgrad_plan")
    nw = 9 * 64 * 128
    slab, dW = R(ns.value, nw), E(nw)
    cases.append((f"slab_sum_vec ({ns.value} slabs, deconv4 dW)", (ns.value + 1) * nw * F,
                  lambda: check(lib.veex_slab_sum(slab.data_ptr(), ns.value, nw, nw, dW.data_ptr(), None, 1, 0, st()))))
    dw6 = E(256)
    cases.append((f"slab_sum_tall ({B} slabs, deconv6 dW)", B * 256 * F,
                  lambda: check(lib.veex_slab_sum(slab6.data_ptr(), B, 448, 256, dw6.data_ptr(), None, 1, 0, st()))))
    sw = E(576)
    cases.append((f"slab_sum_tall ({nb} slabs, conv1 dW)", nb * 576 * F,
                  lambda: check(lib.veex_slab_sum(slab1.data_ptr(), nb, 576, 576, sw.data_ptr(), None, 1, 0, st()))))
    g4, a4, e4, o4 = R(B, 4096), torch.rand(B, 4096, device="cuda"), torch.rand(4096, device="cuda") + 0.5, E(B, 4096)
    chunks = lib.veex_bnrelu_bwd_rows_chunks(B)
    part = E(chunks, 3, 4096)
    keep.extend([a4, e4])
    cases.append(("bnrelu_bwd_rows (dec.dense)", (3 * B * 4096 + chunks * 3 * 4096) * F,
                  lambda: check(lib.veex_bnrelu_bwd_rows(g4.data_ptr(), a4.data_ptr(), e4.data_ptr(), o4.data_ptr(),
                                                         part.data_ptr(), B, 4096, 4096, st()))))
    mean, rstd, db, dg, dbt = R(4096), torch.rand(4096, device="cuda") + 0.5, E(4096), E(4096), E(4096)
    keep.append(rstd)
    cases.append(("colsum_finalize (dec.dense)", (chunks * 3 * 4096 + 6 * 4096) * F,
                  lambda: check(lib.veex_colsum_finalize(part.data_ptr(), chunks, 3 * 4096, 4096, 4096, e4.data_ptr(),
                                                         mean.data_ptr(), rstd.data_ptr(), db.data_ptr(), dg.data_ptr(),
                                                         dbt.data_ptr(), st()))))
    nbn = 5184
    ga, be, mu, va, s1, s2, s3 = R(nbn), R(nbn), R(nbn), torch.rand(nbn, device="cuda") + 0.5, E(nbn), E(nbn), E(nbn)
    keep.append(va)
    cases.append(("bn_prepare", 7 * nbn * F,
                  lambda: check(lib.veex_bn_prepare(ga.data_ptr(), be.data_ptr(), mu.data_ptr(), va.data_ptr(), s1.data_ptr(),
                                                    s2.data_ptr(), s3.data_ptr(), nbn, 1e-3, st()))))
    # either side of the path: normalize / denormalize over the 100k-map set
    big = torch.rand(100_000 * 1600, device="cuda"); keep.append(big)
    out_big = E(big.numel())
    mm = E(2)
    pb = E(2 * lib.veex_minmax_partial_blocks(big.numel()))
    cases.append(("minmax (100k maps)", big.numel() * F,
                  lambda: check(lib.veex_minmax(big.data_ptr(), big.numel(), pb.data_ptr(), mm.data_ptr(), st()))))
    cases.append(("normalize (100k maps)", 2 * big.numel() * F,
                  lambda: check(lib.veex_normalize(big.data_ptr(), mm.data_ptr(), out_big.data_ptr(), big.numel(), st()))))
    cases.append(("denormalize (100k maps)", 2 * big.numel() * F,
                  lambda: check(lib.veex_denormalize(big.data_ptr(), mm.data_ptr(), out_big.data_ptr(), big.numel(), st()))))
    hbm_cases._keep = keep
    return cases


def hbm_table(B=512):
    """Achieved HBM GB/s of every fused elementwise / reduction kernel against the 8 TB/s peak (north_star)."""
    rows = []
    for label, nbytes, fn in hbm_cases(B):
        t = _time_launch(fn, iters=20, warm=3)
        rows.append({"kernel": label, "bytes": nbytes, "us": t * 1e6, "gbs": nbytes / t / 1e9, "frac": nbytes / t / 1e9 / PEAK_HBM_GBS})
    return rows


if __name__ == "__main__":
    import sys
    if "--hbm" in sys.argv:
        args = [a for a in sys.argv[1:] if a != "--hbm"]
        B = int(args[0]) if args else 512
        print(f"HBM-bound kernels at batch {B}: algorithmic bytes (operands once + results once), HIP-event time per launch on the "
              f"launch stream (median of 3 groups of 20 back-to-back launches), against {PEAK_HBM_GBS:.0f} GB/s.\n"
              "Launches under ~10 us are bounded by the 4.6 us begin-to-end floor of any kernel here, not by bytes.")
        for r in hbm_table(B):
            print(f"{r['kernel']:40s} {r['bytes'] / 1e6:9.2f} MB {r['us']:8.1f} us {r['gbs']:8.0f} GB/s  {100 * r['frac']:5.1f}% of HBM peak")
        sys.exit(0)
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    bf = len(sys.argv) > 2 and sys.argv[2] == "bf16"
    x3 = len(sys.argv) > 2 and sys.argv[2] == "f32x3"
    tot = 0.0
    print(f"MFMA launches of one train step at batch {B} ({'bf16' if bf else 'split-fp32' if x3 else 'fp32'} engine), HIP-event time per launch; roof = "
          f"max(FLOP / {'2500' if bf else '416.7 (6 bf16 MFMAs per product)' if x3 else '157.3'} TFLOP/s, algorithmic bytes / 8 TB/s)")
    for r in layer_table(B, bf16=bf, x3=x3):
        tot += r["us"]
        print(f"{r['kernel']:22s} {r['us']:8.1f} us  {r['tflops']:6.1f} TF  {100 * r['frac']:5.1f}% of {'bf16' if bf else 'its' if x3 else 'f32'} MFMA peak"
              f"  bytes {r['bytes'] / 1e6:7.1f} MB  roof {r['roof_us']:6.1f} us ({r['roof']})  {100 * r['frac_roof']:5.1f}% of roof")
    print(f"sum {tot:.1f} us")
