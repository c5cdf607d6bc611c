"""torch-CPU restatement of the reference hot path (TEST INFRASTRUCTURE ONLY; PARITY UNPINNED,
see oracle/__init__.py).

Second, independent formulation of oracle/vae_oracle.py: library convolutions
(F.conv2d with explicit asymmetric F.pad; F.conv_transpose2d padding=0 + crop)
and autograd for every gradient.  In float64 it cross-checks the NumPy oracle; in float32
(oneDNN, all host cores) it is the ``cpu_baseline`` of bench.py (BASELINE.md section 3,
kind "port").
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

from . import vae_oracle as vo


def to_torch(params: dict, dtype=torch.float64, requires_grad=True):
    out = {}
    tr = None
    for k, v in params.items():
        t = torch.tensor(np.asarray(v), dtype=dtype)
        leaf = k.rsplit(".", 1)[1]
        if requires_grad and leaf not in ("mean", "var"):
            t.requires_grad_(True)
        out[k] = t
    return out


def _bn(a, p, prefix, sem=None):
    """BatchNormalization after layer ``prefix``: moving statistics (default) or, with sem bn="batch", the batch's
    own mean and biased variance with autograd flowing through them (vae_oracle.DEFAULT_SEM)."""
    if sem is not None and sem["bn"] == "batch":
        dims = tuple(range(a.dim() - 1))
        mu = a.mean(dim=dims)
        var = a.var(dim=dims, unbiased=False)
        return (a - mu) * torch.rsqrt(var + vo.BN_EPS) * p[prefix + ".gamma"] + p[prefix + ".beta"]
    rstd = torch.rsqrt(p[prefix + ".var"] + vo.BN_EPS)
    return (a - p[prefix + ".mean"]) * rstd * p[prefix + ".gamma"] + p[prefix + ".beta"]


def _conv_same_s2(x_nchw, W, b, stride):
    # TF SAME for k3 s2 on even input: pad 0 before, 1 after
    k = W.shape[0]
    H = x_nchw.shape[2]
    out, pb, pa = vo.same_pad(H, k, stride)
    xp = F.pad(x_nchw, (pb, pa, pb, pa))
    return F.conv2d(xp, W.permute(3, 2, 0, 1), b, stride=stride)


def _deconv(x_nchw, W, b, stride, padding):
    # kernel (kh,kw,Cout,Cin) -> torch conv_transpose2d weight (Cin,Cout,kh,kw)
    k = W.shape[0]
    H = x_nchw.shape[2]
    out, pb, full = vo.deconv_geometry(H, k, stride, padding)
    y = F.conv_transpose2d(x_nchw, W.permute(3, 2, 0, 1), None, stride=stride, padding=0)
    y = y[:, :, pb:pb + out, pb:pb + out]
    return y + b.view(1, -1, 1, 1)


def _relu(pre_nhwc, masks, layer):
    """ReLU; with ``masks`` (layer -> NHWC bool array, the device's decisions) the branch is the mask's, so
    that a comparison with an fp32 implementation is not dominated by the few pre-activations within
    rounding of the kink (tests/util.device_masks asserts that every disagreement is such a kink)."""
    if masks is None or layer not in masks:
        return torch.relu(pre_nhwc)
    m = torch.as_tensor(masks[layer]).reshape(pre_nhwc.shape).to(pre_nhwc.dtype)
    return pre_nhwc * m


def encoder(p, x_nhwc, variant="vae", masks=None, sem=None):
    h = x_nhwc.permute(0, 3, 1, 2)
    acts = {}
    for name, cin, cout, k, s, pad in vo.ENC_CONVS:
        pre = _conv_same_s2(h, p[f"enc.{name}.kernel"], p[f"enc.{name}.bias"], s).permute(0, 2, 3, 1)
        acts[f"enc.{name}.pre"] = pre
        a = _relu(pre, masks, f"enc.{name}")
        acts[f"enc.{name}.a"] = a
        h = _bn(a, p, f"enc.{name}", sem).permute(0, 3, 1, 2)
    flat = h.permute(0, 2, 3, 1).reshape(h.shape[0], -1)
    pre = flat @ p["enc.dense.kernel"] + p["enc.dense.bias"]
    acts["enc.dense.pre"] = pre
    d = _relu(pre, masks, "enc.dense")
    acts["enc.dense.a"] = d
    if variant == "vae":
        zm = d @ p["enc.z_mean.kernel"] + p["enc.z_mean.bias"]
        zlv = d @ p["enc.z_log_var.kernel"] + p["enc.z_log_var.bias"]
        return zm, zlv, acts
    return d, None, acts


def decoder(p, z, masks=None, acts=None, sem=None):
    pre = z @ p["dec.dense.kernel"] + p["dec.dense.bias"]
    if acts is not None:
        acts["dec.dense.pre"] = pre
    a = _relu(pre, masks, "dec.dense")
    h = _bn(a, p, "dec.dense", sem).reshape(-1, 4, 4, 256).permute(0, 3, 1, 2)
    for name, cin, cout, k, s, pad in vo.DEC_DECONVS:
        pre = _deconv(h, p[f"dec.{name}.kernel"], p[f"dec.{name}.bias"], s, pad).permute(0, 2, 3, 1)
        if name == "deconv6":
            return pre                                 # logits NHWC
        if acts is not None:
            acts[f"dec.{name}.pre"] = pre
        a = _relu(pre, masks, f"dec.{name}")
        h = _bn(a, p, f"dec.{name}", sem).permute(0, 3, 1, 2)


def losses(p, x, eps=None, variant="vae", masks=None, acts=None, sem=None):
    sem = vo.resolve_sem(sem)
    zm, zlv, enc_acts = encoder(p, x, variant, masks, sem)
    if acts is not None:
        acts.update(enc_acts)
    if variant == "vae":
        z = zm + torch.exp(0.5 * zlv) * eps
    else:
        z = zm
    logits = decoder(p, z, masks, acts, sem)
    if sem["bce"] == "probs":
        raise ValueError('torch_ref evaluates in float64: ask for bce="probs64" (vae_oracle.bce_probs, f32=False)')
    if sem["bce"] == "probs64":    # clipped-probability form; torch.clamp passes no gradient outside its range, like TF
        pr = torch.clamp(torch.sigmoid(logits), vo.BCE_CLIP, vo.BCE_CLIP_HI)
        bce = -(x * torch.log(pr + vo.BCE_CLIP) + (1.0 - x) * torch.log(1.0 - pr + vo.BCE_CLIP))
    else:
        bce = torch.clamp(logits, min=0) - logits * x + torch.log1p(torch.exp(-logits.abs()))
    rec = bce.mean(dim=-1).sum(dim=(1, 2)).mean()
    if variant == "vae":
        kl = (-0.5 * (1 + zlv - zm ** 2 - torch.exp(zlv))).sum(dim=1).mean()
    else:
        kl = torch.zeros((), dtype=x.dtype)
    return rec + kl, rec, kl, logits


def pre_activations(params_np: dict, x_np, eps_np=None, variant="vae", dtype=torch.float64, sem=None):
    """{layer + ".pre": ndarray} of every ReLU layer (what tests/util.device_masks compares the device's
    ReLU decisions with), plus "logits"."""
    p = to_torch(params_np, dtype, requires_grad=False)
    x = torch.tensor(np.asarray(x_np), dtype=dtype)
    eps = None if eps_np is None else torch.tensor(np.asarray(eps_np), dtype=dtype)
    acts = {}
    with torch.no_grad():
        total, rec, kl, logits = losses(p, x, eps, variant, None, acts, sem)
    out = {k: v.numpy() for k, v in acts.items() if k.endswith(".pre")}
    out["logits"] = logits.numpy()
    return out


def grads(params_np: dict, x_np, eps_np=None, variant="vae", dtype=torch.float64, masks=None, sem=None):
    """Returns (total, rec, kl, {name: grad ndarray}) via autograd."""
    p = to_torch(params_np, dtype)
    x = torch.tensor(np.asarray(x_np), dtype=dtype)
    eps = None if eps_np is None else torch.tensor(np.asarray(eps_np), dtype=dtype)
    total, rec, kl, logits = losses(p, x, eps, variant, masks, sem=sem)
    names = [k for k, t in p.items() if t.requires_grad]
    gs = torch.autograd.grad(total, [p[k] for k in names])
    return float(total.detach()), float(rec.detach()), float(kl.detach()), {k: g.numpy() for k, g in zip(names, gs)}


def saliency(params_np, x_np, units, variant="vae", dtype=torch.float64, grad_x_input=False, masks=None):
    p = to_torch(params_np, dtype, requires_grad=False)
    x = torch.tensor(np.asarray(x_np), dtype=dtype, requires_grad=True)
    zm, _, _ = encoder(p, x, variant, masks)
    out = []
    for j in units:
        (g,) = torch.autograd.grad(zm[:, j].sum(), x, retain_graph=True)
        out.append((g * x.detach() if grad_x_input else g).numpy())
    return np.stack(out)


class TorchTrainer:
    """fp32 CPU trainer with TF-form Adam -- the timed ``cpu_baseline`` ("port")."""

    def __init__(self, params_np, lr, variant="vae", dtype=torch.float32):
        self.p = to_torch(params_np, dtype)
        self.names = [k for k, t in self.p.items() if t.requires_grad]
        self.m = {k: torch.zeros_like(self.p[k]) for k in self.names}
        self.v = {k: torch.zeros_like(self.p[k]) for k in self.names}
        self.t = 0
        self.lr = lr
        self.variant = variant
        self.dtype = dtype

    def train_step(self, x_np, eps_np=None):
        x = torch.as_tensor(np.asarray(x_np), dtype=self.dtype)
        if self.variant == "vae":
            if eps_np is None:
                eps = torch.randn(x.shape[0], self.p["dec.dense.kernel"].shape[0], dtype=self.dtype)
            else:
                eps = torch.as_tensor(np.asarray(eps_np), dtype=self.dtype)
        else:
            eps = None
        total, rec, kl, _ = losses(self.p, x, eps, self.variant)
        gs = torch.autograd.grad(total, [self.p[k] for k in self.names])
        self.t += 1
        alpha = self.lr * math.sqrt(1 - vo.ADAM_B2 ** self.t) / (1 - vo.ADAM_B1 ** self.t)
        with torch.no_grad():
            for k, g in zip(self.names, gs):
                self.m[k].mul_(vo.ADAM_B1).add_(g, alpha=1 - vo.ADAM_B1)
                self.v[k].mul_(vo.ADAM_B2).addcmul_(g, g, value=1 - vo.ADAM_B2)
                self.p[k].sub_(alpha * self.m[k] / (self.v[k].sqrt() + vo.ADAM_EPS))
        return float(total.detach()), float(rec.detach()), float(kl.detach())

    def params_numpy(self):
        return {k: t.detach().numpy().copy() for k, t in self.p.items()}
