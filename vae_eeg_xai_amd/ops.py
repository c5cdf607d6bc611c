"""PyTorch-ROCm custom ops over the C-ABI: ``torch.ops.veex.*``.

Every op body is one ``ctypes`` call into libveex.so (include/veex.h) with raw device pointers and
the current HIP stream; torch contributes memory, streams and the dispatcher, no arithmetic.  The ops
are registered for the ``cuda`` device type ONLY: there is no CPU kernel, so a CPU tensor raises
NotImplementedError in the dispatcher (the product has no CPU path).  ``register_fake`` gives each op
its output shapes, so the ops trace (FakeTensor / torch.export) without running.

Autograd.  The library's backward is the fused whole-step ``veex::train_fwd_bwd`` (VAE.train_step up
to tape.gradient, models.py:148-166): every weight gradient lands in the flat ``grads`` buffer in one
call, so there are no per-layer VJP symbols to hang on ``register_autograd`` -- except ``sample``
(models.py:10-22), whose forward and backward kernels exist on their own (veex_reparam_kl_fwd/bwd):
``veex::sample`` is differentiable in ``z_mean`` and ``z_log_var``.

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
@torch.library.custom_op("veex::train_fwd_bwd", mutates_args=("ws", "grads", "metrics", "eps"), device_types=_CUDA)
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

__all__ = ["train_fwd_bwd", "forward", "encode", "decode", "attribute", "adam", "sample", "sample_bwd"]
