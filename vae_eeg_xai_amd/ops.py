"""PyTorch-ROCm custom ops over the C-ABI: ``torch.ops.veex.*``.

Every op body is one ``ctypes`` call into libveex.so (include/veex.h) with raw device pointers and
the current HIP stream; torch contributes memory, streams and the dispatcher, no arithmetic.  The ops
are registered for the ``cuda`` device type ONLY: there is no CPU kernel, so a CPU tensor raises
NotImplementedError in the dispatcher (the product has no CPU path).  ``register_fake`` gives each op
its output shapes, so the ops trace (FakeTensor / torch.export) without running.

Autograd.  The product's backward is the fused whole-step ``veex::train_fwd_bwd`` (VAE.train_step up
to tape.gradient, models.py:148-166): every weight gradient lands in the flat ``grads`` buffer in one
call.  Beside it, the layer-level ops ``veex::conv2d`` and ``veex::conv_transpose2d`` (the Conv2D /
Conv2DTranspose(+ReLU) layers of models.py:219-298 with Keras 'same' / 'valid' geometry) are
differentiable through ``register_autograd``: their backward maps onto the operator-level symbols the
whole-step path is built from -- veex_bnrelu_bwd_rows (ReLU mask + bias gradient), veex_gather_gemm in
the dual mode (input gradient), veex_wgrad_gemm + veex_slab_sum (weight gradient) -- so a PyTorch user
can compose them under torch.autograd; ``veex::sample`` (models.py:10-22) likewise
(veex_reparam_kl_fwd/bwd), differentiable in ``z_mean`` and ``z_log_var``.

A ``VeexNet*`` travels as a Python int (the plan is host-only state; it owns no device memory).
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Tuple

import torch
from torch import Tensor

from . import _lib
from ._lib import lib, check, ptr

_CUDA = "cuda"


def _h(net: int):
    return C.c_void_p(net)


# ------------------------------------------------------------------ whole-step ops
# (state: written only under set_semantics(bn="batch"), which updates the BatchNorm moving statistics)
@torch.library.custom_op("veex::train_fwd_bwd", mutates_args=("ws", "grads", "metrics", "eps", "state"), device_types=_CUDA)
def train_fwd_bwd(net: int, params: Tensor, state: Tensor, x: Tensor, eps: Optional[Tensor], flags: int, seed: int,
                  first_sample: int, ws: Tensor, grads: Tensor, metrics: Optional[Tensor]) -> None:
    """VAE.train_step up to tape.gradient (models.py:148-166): all gradients into ``grads[:n_params]``,
    (total, rec, kl) into ``grads[n_params:n_params+3]``."""
    n = lib.veex_net_param_floats(_h(net))
    losses = grads[n:n + 3]
    check(lib.veex_net_train_fwd_bwd(_h(net), ptr(params), ptr(state), ptr(x), ptr(eps), x.shape[0], flags, seed,
                                     first_sample, ptr(ws), ws.numel(), ptr(grads), ptr(losses), ptr(metrics),
                                     _lib.current_stream()), "net_train_fwd_bwd")


@torch.library.custom_op("veex::forward", mutates_args=("ws", "metrics", "eps"), device_types=_CUDA)
def forward(net: int, params: Tensor, state: Tensor, x: Tensor, eps: Optional[Tensor], flags: int, seed: int,
            first_sample: int, ws: Tensor, metrics: Optional[Tensor], want_recon: bool) -> Tuple[Tensor, Tensor]:
    """VAE.call / test_step (models.py:135-138,182-207) -> (recon (B,40,40,1) or empty, losses[3])."""
    B = x.shape[0]
    recon = torch.empty((B, 40, 40, 1) if want_recon else (0,), dtype=torch.float32, device=x.device)
    losses = torch.empty(3, dtype=torch.float32, device=x.device)
    check(lib.veex_net_forward(_h(net), ptr(params), ptr(state), ptr(x), ptr(eps), B, flags, seed, first_sample,
                               ptr(ws), ws.numel(), ptr(recon) if want_recon else None, ptr(losses), ptr(metrics),
                               _lib.current_stream()), "net_forward")
    return recon, losses


@forward.register_fake
def _(net, params, state, x, eps, flags, seed, first_sample, ws, metrics, want_recon):
    B = x.shape[0]
    return (x.new_empty((B, 40, 40, 1) if want_recon else (0,), dtype=torch.float32),
            x.new_empty(3, dtype=torch.float32))


@torch.library.custom_op("veex::encode", mutates_args=("ws", "eps"), device_types=_CUDA)
def encode(net: int, params: Tensor, state: Tensor, x: Tensor, eps: Optional[Tensor], flags: int, seed: int,
           first_sample: int, ws: Tensor, latent: int, vae: bool) -> Tuple[Tensor, Tensor, Tensor]:
    """Encoder.call (models.py:244-253) -> (z_mean, z_log_var, z); the plain autoencoder
    (models_standard.py:110-116) fills only z and returns empty tensors for the other two."""
    B = x.shape[0]
    z = torch.empty(B, latent, dtype=torch.float32, device=x.device)
    zm = torch.empty_like(z) if vae else z.new_empty(0)
    zlv = torch.empty_like(z) if vae else z.new_empty(0)
    check(lib.veex_net_encode(_h(net), ptr(params), ptr(state), ptr(x), ptr(eps), B, flags, seed, first_sample,
                              ptr(ws), ws.numel(), ptr(zm) if vae else None, ptr(zlv) if vae else None, ptr(z),
                              _lib.current_stream()), "net_encode")
    return zm, zlv, z


@encode.register_fake
def _(net, params, state, x, eps, flags, seed, first_sample, ws, latent, vae):
    B = x.shape[0]
    z = x.new_empty((B, latent), dtype=torch.float32)
    e = x.new_empty((B, latent) if vae else (0,), dtype=torch.float32)
    return e, torch.empty_like(e), z


@torch.library.custom_op("veex::decode", mutates_args=("ws",), device_types=_CUDA)
def decode(net: int, params: Tensor, state: Tensor, z: Tensor, ws: Tensor, want_logits: bool) -> Tuple[Tensor, Tensor]:
    """Decoder.call (models.py:303-312) -> (recon (B,40,40,1), logits or empty)."""
    B = z.shape[0]
    recon = torch.empty(B, 40, 40, 1, dtype=torch.float32, device=z.device)
    logits = torch.empty_like(recon) if want_logits else recon.new_empty(0)
    check(lib.veex_net_decode(_h(net), ptr(params), ptr(state), ptr(z), B, ptr(ws), ws.numel(), ptr(recon),
                              ptr(logits) if want_logits else None, _lib.current_stream()), "net_decode")
    return recon, logits


@decode.register_fake
def _(net, params, state, z, ws, want_logits):
    B = z.shape[0]
    r = z.new_empty((B, 40, 40, 1), dtype=torch.float32)
    return r, (torch.empty_like(r) if want_logits else z.new_empty(0, dtype=torch.float32))


@torch.library.custom_op("veex::attribute", mutates_args=("ws",), device_types=_CUDA)
def attribute(net: int, params: Tensor, state: Tensor, x: Tensor, units: List[int], grad_x_input: bool,
              ws: Tensor) -> Tensor:
    """Saliency / gradient*input maps d z_mean[:, j] / d x -> (len(units), B, 40, 40)
    (north_star item 3; no reference call site)."""
    B = x.shape[0]
    out = torch.empty(len(units), B, 40, 40, dtype=torch.float32, device=x.device)
    arr = (C.c_int * len(units))(*units)
    check(lib.veex_net_attribute(_h(net), ptr(params), ptr(state), ptr(x), B, arr, len(units),
                                 1 if grad_x_input else 0, ptr(ws), ws.numel(), ptr(out), _lib.current_stream()),
          "net_attribute")
    return out


@attribute.register_fake
def _(net, params, state, x, units, grad_x_input, ws):
    return x.new_empty((len(units), x.shape[0], 40, 40), dtype=torch.float32)


@torch.library.custom_op("veex::adam", mutates_args=("p", "m", "v", "step", "alpha"), device_types=_CUDA)
def adam(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: Tensor, alpha: Tensor, lr: float, gscale: float,
         b1: float, b2: float, eps: float) -> None:
    """keras.optimizers.legacy.Adam on the flat buffers (train.py:9,137; models.py:169)."""
    check(lib.veex_adam(ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), ptr(step), ptr(alpha), lr, gscale, b1, b2, eps,
                        _lib.current_stream()), "adam")


# ------------------------------------------------------------------ sample(): the one op with its own backward symbol
@torch.library.custom_op("veex::sample", mutates_args=(), device_types=_CUDA)
def sample(z_mean: Tensor, z_log_var: Tensor, eps: Tensor) -> Tensor:
    """models.py:10-22 with an injected epsilon: z = z_mean + exp(0.5 * z_log_var) * eps."""
    B, L = z_mean.shape
    mulv = torch.cat([z_mean, z_log_var], dim=1).contiguous().float()
    e = eps.contiguous().float()
    z = torch.empty(B, L, dtype=torch.float32, device=z_mean.device)
    check(lib.veex_reparam_kl_fwd(ptr(mulv), ptr(e), ptr(z), None, B, L, 0, 0, 0, _lib.current_stream()), "reparam_kl_fwd")
    return z


@sample.register_fake
def _(z_mean, z_log_var, eps):
    return z_mean.new_empty(z_mean.shape, dtype=torch.float32)


@torch.library.custom_op("veex::sample_bwd", mutates_args=(), device_types=_CUDA)
def sample_bwd(dz: Tensor, z_mean: Tensor, z_log_var: Tensor, eps: Tensor) -> Tuple[Tensor, Tensor]:
    """VJP of veex::sample: (dz, dz * 0.5 * exp(0.5 * z_log_var) * eps) -- veex_reparam_kl_bwd without its KL
    term (inv_batch = 0)."""
    B, L = z_mean.shape
    mulv = torch.cat([z_mean, z_log_var], dim=1).contiguous().float()
    dmulv = torch.empty_like(mulv)
    check(lib.veex_reparam_kl_bwd(ptr(dz.contiguous().float()), ptr(mulv), ptr(eps.contiguous().float()), ptr(dmulv),
                                  B, L, 0.0, _lib.current_stream()), "reparam_kl_bwd")
    return dmulv[:, :L].contiguous(), dmulv[:, L:].contiguous()


@sample_bwd.register_fake
def _(dz, z_mean, z_log_var, eps):
    return torch.empty_like(z_mean), torch.empty_like(z_mean)


def _sample_setup(ctx, inputs, output):
    ctx.save_for_backward(*inputs)


def _sample_backward(ctx, dz):
    z_mean, z_log_var, eps = ctx.saved_tensors
    dmu, dlv = torch.ops.veex.sample_bwd(dz, z_mean, z_log_var, eps)
    return dmu, dlv, None


sample.register_autograd(_sample_backward, setup_context=_sample_setup)



# ------------------------------------------------------------------ layer-level ops with autograd
def conv_geometry(n_in: int, k: int, stride: int, padding: str) -> Tuple[int, int]:
    """Conv2D output length and pad-before, TF arithmetic ('same' pads total//2 before)."""
    if padding == "same":
        out = -(-n_in // stride)
        tot = max((out - 1) * stride + k - n_in, 0)
        return out, tot // 2
    if padding != "valid":
        raise ValueError("padding must be 'same' or 'valid'")
    return (n_in - k) // stride + 1, 0


def deconv_geometry(n_in: int, k: int, stride: int, padding: str) -> Tuple[int, int]:
    """Conv2DTranspose output length and crop-at-start, Keras arithmetic."""
    if padding == "valid":
        return n_in * stride + max(k - stride, 0), 0
    if padding != "same":
        raise ValueError("padding must be 'same' or 'valid'")
    out = n_in * stride
    return out, conv_geometry(out, k, stride, "same")[1]


def _check_layer_args(x: Tensor, w: Tensor, stride: int):
    if x.dim() != 4 or w.dim() != 4 or x.dtype != torch.float32 or w.dtype != torch.float32:
        raise ValueError("x: (N,H,W,C) float32 NHWC, w: 4-d float32")
    if stride not in (1, 2):
        raise ValueError("stride must be 1 or 2")
    if x.shape[1] != x.shape[2] or w.shape[0] != w.shape[1]:
        raise ValueError("square maps and kernels only (the reference has no others)")


def _gemm(mode: int, a: Tensor, w: Tensor, bias: Optional[Tensor], out: Tensor, cin: int, cout: int, k: int,
          stride: int, pad_begin: int, relu: bool, what: str) -> None:
    d = _lib.GemmDesc()
    d.struct_size = C.sizeof(_lib.GemmDesc)
    d.inp, d.weight, d.out, d.bias = ptr(a), ptr(w), ptr(out), ptr(bias)
    d.N, d.Hin, d.Win, d.Cin = a.shape[0], a.shape[1], a.shape[2], cin
    d.Hout, d.Wout, d.Cout = out.shape[1], out.shape[2], cout
    d.kh = d.kw = k
    d.stride, d.pad_begin, d.mode, d.epilogue, d.relu, d.nsplit = stride, pad_begin, mode, 0, 1 if relu else 0, 1
    check(lib.veex_gather_gemm(C.byref(d), _lib.current_stream()), what)


def _relu_bias_bwd(dy: Tensor, y: Tensor, relu: bool) -> Tuple[Tensor, Tensor]:
    """d(pre-activation) = dy * (y > 0) and the bias gradient = its sum over (N,H,W), both by library kernels
    (the row-wise ReLU backward of dec.dense, run over N*H*W rows of Cout columns)."""
    cout = dy.shape[-1]
    rows = dy.numel() // cout
    chunks = lib.veex_bnrelu_bwd_rows_chunks(rows)
    part = torch.empty(chunks * 3 * cout, dtype=torch.float32, device=dy.device)
    dpre = torch.empty_like(dy)
    g2, y2, o2 = dy.view(rows, cout), y.view(rows, cout), dpre.view(rows, cout)
    step = 1 << 19                                   # rows per call (the kernel's grid limit is 65535 chunks of 8 rows)
    for r0 in range(0, rows, step):
        r1 = min(rows, r0 + step)
        c0 = lib.veex_bnrelu_bwd_rows_chunks(r0)     # r0 is a multiple of the chunk height: chunks stay contiguous
        check(lib.veex_bnrelu_bwd_rows(ptr(g2[r0:r1]), ptr(y2[r0:r1]) if relu else None, None, ptr(o2[r0:r1]),
                                       ptr(part[c0 * 3 * cout:]), r1 - r0, cout, 0, _lib.current_stream()), "relu backward")
    db = torch.empty(cout, dtype=torch.float32, device=dy.device)
    check(lib.veex_colsum_finalize(ptr(part), chunks, 3 * cout, cout, cout, None, None, None, ptr(db), None, None,
                                   _lib.current_stream()), "bias gradient")
    return dpre, db


def _wgrad(big: Tensor, small: Tensor, k: int, stride: int, pad_begin: int) -> Tensor:
    """dW[tap][cb][cs] = sum over pixels of big(shifted by the tap) * small (veex_wgrad_gemm + veex_slab_sum)."""
    w = _lib.WgradDesc()
    w.big, w.small = ptr(big), ptr(small)
    w.N, w.Hb, w.Wb, w.Cb = big.shape[0], big.shape[1], big.shape[2], big.shape[3]
    w.Hs, w.Ws, w.Cs = small.shape[1], small.shape[2], small.shape[3]
    w.kh = w.kw = k
    w.stride, w.pad_begin = stride, pad_begin
    ns, per = C.c_int32(), C.c_int32()
    check(lib.veex_wgrad_plan(C.byref(w), C.byref(ns), C.byref(per)), "wgrad_plan")
    w.nsplit, w.rows_per_split = ns.value, per.value
    n = k * k * big.shape[3] * small.shape[3]
    dw = torch.empty(k, k, big.shape[3], small.shape[3], dtype=torch.float32, device=big.device)
    if ns.value == 1:
        w.out = ptr(dw)
        check(lib.veex_wgrad_gemm(C.byref(w), _lib.current_stream()), "wgrad")
        return dw
    slab = torch.empty(ns.value * n, dtype=torch.float32, device=big.device)
    w.out = ptr(slab)
    check(lib.veex_wgrad_gemm(C.byref(w), _lib.current_stream()), "wgrad")
    check(lib.veex_slab_sum(ptr(slab), ns.value, n, n, ptr(dw), None, 1, 0, _lib.current_stream()), "slab_sum")
    return dw


@torch.library.custom_op("veex::conv2d", mutates_args=(), device_types=_CUDA)
def conv2d(x: Tensor, w: Tensor, bias: Tensor, stride: int, padding: str, relu: bool) -> Tensor:
    """keras.layers.Conv2D(Cout, k, strides, padding, activation='relu' or None) on NHWC float32
    (models.py:219-235): x (N,H,W,Cin), w (k,k,Cin,Cout), bias (Cout) -> (N,Ho,Wo,Cout)."""
    _check_layer_args(x, w, stride)
    k, cin, cout = w.shape[0], w.shape[2], w.shape[3]
    ho, pb = conv_geometry(x.shape[1], k, stride, padding)
    x, w, bias = x.contiguous(), w.contiguous(), bias.contiguous()
    y = torch.empty(x.shape[0], ho, ho, cout, dtype=torch.float32, device=x.device)
    _gemm(0, x, w, bias, y, cin, cout, k, stride, pb, relu, "conv2d")
    return y


@conv2d.register_fake
def _(x, w, bias, stride, padding, relu):
    ho, _ = conv_geometry(x.shape[1], w.shape[0], stride, padding)
    return x.new_empty((x.shape[0], ho, ho, w.shape[3]), dtype=torch.float32)


@torch.library.custom_op("veex::conv2d_bwd", mutates_args=(), device_types=_CUDA)
def conv2d_bwd(dy: Tensor, x: Tensor, w: Tensor, y: Tensor, stride: int, padding: str, relu: bool) -> Tuple[Tensor, Tensor, Tensor]:
    """VJP of veex::conv2d -> (dx, dw, dbias): ReLU mask + bias gradient, input gradient as the dual
    transposed-convolution gather (mode 1), weight gradient with big = x, small = d(pre-activation)."""
    k, cin, cout = w.shape[0], w.shape[2], w.shape[3]
    _, pb = conv_geometry(x.shape[1], k, stride, padding)
    x, w = x.contiguous(), w.contiguous()
    dpre, db = _relu_bias_bwd(dy.contiguous(), y.contiguous(), relu)
    dx = torch.empty_like(x)
    _gemm(1, dpre, w, None, dx, cout, cin, k, stride, pb, False, "conv2d input gradient")
    dw = _wgrad(x, dpre, k, stride, pb)
    return dx, dw, db


@conv2d_bwd.register_fake
def _(dy, x, w, y, stride, padding, relu):
    return torch.empty_like(x), torch.empty_like(w), w.new_empty(w.shape[3])


def _conv2d_setup(ctx, inputs, output):
    x, w, bias, stride, padding, relu = inputs
    ctx.save_for_backward(x, w, output)
    ctx.args = (stride, padding, relu)


def _conv2d_backward(ctx, dy):
    x, w, y = ctx.saved_tensors
    dx, dw, db = torch.ops.veex.conv2d_bwd(dy, x, w, y, *ctx.args)
    return dx, dw, db, None, None, None


conv2d.register_autograd(_conv2d_backward, setup_context=_conv2d_setup)


@torch.library.custom_op("veex::conv_transpose2d", mutates_args=(), device_types=_CUDA)
def conv_transpose2d(x: Tensor, w: Tensor, bias: Tensor, stride: int, padding: str, relu: bool) -> Tensor:
    """keras.layers.Conv2DTranspose(Cout, k, strides, padding, activation='relu' or None) on NHWC float32
    (models.py:270-298): x (N,H,W,Cin), w (k,k,Cout,Cin), bias (Cout) -> (N,Ho,Wo,Cout); 'same' crops at the end."""
    _check_layer_args(x, w, stride)
    k, cout, cin = w.shape[0], w.shape[2], w.shape[3]
    ho, pb = deconv_geometry(x.shape[1], k, stride, padding)
    x, w, bias = x.contiguous(), w.contiguous(), bias.contiguous()
    y = torch.empty(x.shape[0], ho, ho, cout, dtype=torch.float32, device=x.device)
    _gemm(1, x, w, bias, y, cin, cout, k, stride, pb, relu, "conv_transpose2d")
    return y


@conv_transpose2d.register_fake
def _(x, w, bias, stride, padding, relu):
    ho, _ = deconv_geometry(x.shape[1], w.shape[0], stride, padding)
    return x.new_empty((x.shape[0], ho, ho, w.shape[2]), dtype=torch.float32)


@torch.library.custom_op("veex::conv_transpose2d_bwd", mutates_args=(), device_types=_CUDA)
def conv_transpose2d_bwd(dy: Tensor, x: Tensor, w: Tensor, y: Tensor, stride: int, padding: str,
                         relu: bool) -> Tuple[Tensor, Tensor, Tensor]:
    """VJP of veex::conv_transpose2d -> (dx, dw, dbias): input gradient as the dual convolution gather (mode 0),
    weight gradient with big = d(pre-activation), small = x."""
    k, cout, cin = w.shape[0], w.shape[2], w.shape[3]
    _, pb = deconv_geometry(x.shape[1], k, stride, padding)
    x, w = x.contiguous(), w.contiguous()
    dpre, db = _relu_bias_bwd(dy.contiguous(), y.contiguous(), relu)
    dx = torch.empty_like(x)
    _gemm(0, dpre, w, None, dx, cout, cin, k, stride, pb, False, "conv_transpose2d input gradient")
    dw = _wgrad(dpre, x, k, stride, pb)
    return dx, dw, db


@conv_transpose2d_bwd.register_fake
def _(dy, x, w, y, stride, padding, relu):
    return torch.empty_like(x), torch.empty_like(w), w.new_empty(w.shape[2])


def _deconv_setup(ctx, inputs, output):
    x, w, bias, stride, padding, relu = inputs
    ctx.save_for_backward(x, w, output)
    ctx.args = (stride, padding, relu)


def _deconv_backward(ctx, dy):
    x, w, y = ctx.saved_tensors
    dx, dw, db = torch.ops.veex.conv_transpose2d_bwd(dy, x, w, y, *ctx.args)
    return dx, dw, db, None, None, None


conv_transpose2d.register_autograd(_deconv_backward, setup_context=_deconv_setup)

__all__ = ["train_fwd_bwd", "forward", "encode", "decode", "attribute", "adam", "sample", "sample_bwd",
           "conv2d", "conv2d_bwd", "conv_transpose2d", "conv_transpose2d_bwd", "conv_geometry", "deconv_geometry"]
