"""Device-resident state of one reference network and the whole-step calls into libveex.so.

``Net`` owns the flat fp32 buffers in HBM (trainable parameters, gradients, Adam moments,
BatchNorm moving statistics, workspace) as torch tensors and hands raw pointers + the current
HIP stream to the C-ABI (include/veex.h) through the custom ops of ops.py (torch.ops.veex.*).  torch is
plumbing only (memory, streams, dispatcher, torch.distributed); every arithmetic op of the hot path
runs in the HIP library.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Iterable, List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from . import ops as _ops  # noqa: F401  (registers torch.ops.veex.*)
from ._lib import lib, check, ptr

ADAM_B1, ADAM_B2, ADAM_EPS = 0.9, 0.999, 1e-7   # keras legacy Adam defaults (train.py:9,137)


class TensorSpec:
    __slots__ = ("name", "offset", "shape", "trainable", "part")

    def __init__(self, name, offset, shape, trainable, part):
        self.name, self.offset, self.shape, self.trainable, self.part = name, offset, shape, trainable, part


def _as_device_f32(a, device) -> torch.Tensor:
    if isinstance(a, torch.Tensor):
        t = a.detach()
    elif hasattr(a, "_t") and isinstance(getattr(a, "_t"), torch.Tensor):   # DeviceArray
        t = a._t
    else:
        t = torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float32)))
    return t.to(device=device, dtype=torch.float32).contiguous()


class Net:
    """One VAE (variant 'vae', models.py) or plain autoencoder ('ae', models_standard.py)."""

    def __init__(self, latent_dim: int, variant: str = "vae", device: Optional[str] = None, compute: str = "f32",
                 overlap: bool = True):
        if not torch.cuda.is_available():
            raise _lib.VeexError("no GPU visible: vae_eeg_xai_amd has no CPU path")
        self.device = torch.device(device or f"cuda:{torch.cuda.current_device()}")
        self.latent_dim = int(latent_dim)
        self.variant = variant
        self._h = lib.veex_net_create(self.latent_dim, _lib.VARIANT_VAE if variant == "vae" else _lib.VARIANT_AE)
        if not self._h:
            raise _lib.VeexError(lib.veex_last_error().decode())
        self.compute = "f32"
        self._ws: Dict[int, torch.Tensor] = {}
        self.set_compute(compute)
        self._side_stream = None
        self.set_overlap(overlap)
        self.n_params = lib.veex_net_param_floats(self._h)
        self.n_state = lib.veex_net_state_floats(self._h)
        self.specs: List[TensorSpec] = []
        name = C.create_string_buffer(128)
        off, nd, tr, part = C.c_longlong(), C.c_int(), C.c_int(), C.c_int()
        shp = (C.c_longlong * 4)()
        for i in range(lib.veex_net_num_tensors(self._h)):
            check(lib.veex_net_tensor_info(self._h, i, name, 128, C.byref(off), C.byref(nd), shp, C.byref(tr),
                                           C.byref(part)), "tensor_info")
            self.specs.append(TensorSpec(name.value.decode(), off.value, tuple(shp[k] for k in range(nd.value)),
                                         bool(tr.value), part.value))
        with torch.cuda.device(self.device):
            # 4 extra floats at the tail of the gradient buffer carry (total, rec, kl, pad) so that
            # data-parallel training needs exactly one all-reduce per step
            self.params = torch.zeros(self.n_params, dtype=torch.float32, device=self.device)
            self.grads = torch.zeros(self.n_params + 4, dtype=torch.float32, device=self.device)
            self.adam_m = torch.zeros(self.n_params, dtype=torch.float32, device=self.device)
            self.adam_v = torch.zeros(self.n_params, dtype=torch.float32, device=self.device)
            self.state = torch.zeros(self.n_state, dtype=torch.float32, device=self.device)
            self.adam_step_count = torch.zeros(1, dtype=torch.int32, device=self.device)
            self.adam_alpha = torch.zeros(1, dtype=torch.float32, device=self.device)
            self.metrics = torch.zeros(4, dtype=torch.float32, device=self.device)
        self.philox_seed = 42
        self.samples_seen = 0
        self.semantics = {"bn": "inference", "bce": "logits"}
        # the whole-step calls go through torch.ops.veex.* (ops.py: custom ops whose bodies are the ctypes
        # calls); ``net.via_ops = False`` binds the C-ABI directly (same kernels; for measuring the dispatcher)
        self.via_ops = True
        self.reset_bn_state()

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            lib.veex_net_destroy(h)
            self._h = None

    def set_compute(self, compute: str):
        """Arithmetic of the seven MFMA-bound conv / deconv layers (forward, input and weight
        gradients): "f32" = exact fp32 MFMA (parity path), "bf16" = bf16 MFMA with fp32 accumulation,
        fp32 activations in HBM and fp32 master weights; "f32x3" = fp32 results on the bf16 matrix cores (operands
        split into three bf16 terms, six partial products per multiply, fp32 accumulate: the same 1e-5 parity bar as
        "f32", csrc/gather_gemm_x3.hip).  Layers too small for a 128-row tile (N*Hout*Wout < 4096) stay on the fp32
        kernels."""
        codes = {"f32": _lib.COMPUTE_F32, "bf16": _lib.COMPUTE_BF16, "f32x3": _lib.COMPUTE_F32X3}
        if compute not in codes:
            raise ValueError("compute must be 'f32', 'f32x3' or 'bf16'")
        check(lib.veex_net_set_compute(self._h, codes[compute]), "net_set_compute")
        self.compute = compute
        self._drop_workspaces()     # the workspace plan (split-K slabs) depends on the kernel family

    def set_semantics(self, bn: str = "inference", bce: str = "logits"):
        """The two Keras 2.13 readings of models.py:152-160 that nothing in the reference tree can pin (SURVEY A.3 /
        A.4), each one switch away from its alternative (include/veex.h veex_net_set_semantics).  Defaults = the
        survey's reading: BatchNormalization on its moving statistics inside train_step, BCE from logits.
        ``bn="batch"``: batch statistics + gradients through them + moving-statistics update (fp32 engine);
        ``bce="probs"``: keras.backend.binary_crossentropy on the clipped probabilities."""
        if bn not in ("inference", "batch") or bce not in ("logits", "probs"):
            raise ValueError("bn: 'inference' | 'batch', bce: 'logits' | 'probs'")
        check(lib.veex_net_set_semantics(self._h, 1 if bn == "batch" else 0, 1 if bce == "probs" else 0), "net_set_semantics")
        self.semantics = {"bn": bn, "bce": bce}
        self._drop_graphs()

    def set_overlap(self, on: bool):
        """Run the weight-gradient GEMMs of the backward pass on a second HIP stream, concurrently
        with the input-gradient chain (joined before train_fwd_bwd returns).  VEEX_OVERLAP=0 forces one stream
        (profiling: kernel durations then are stand-alone)."""
        on = bool(on) and os.environ.get("VEEX_OVERLAP", "1") != "0"
        if on and self._side_stream is None:
            with torch.cuda.device(self.device):
                # (normal priority on purpose: a LOWER-priority side stream measured -1.4 .. -2.0 % on the step, and then hung
                #  a test in which a stream of normal priority waited on it -- DESIGN.md 7)
                self._side_stream = torch.cuda.Stream(device=self.device, priority=0)
        s = self._side_stream.cuda_stream if on else None
        check(lib.veex_net_set_side_stream(self._h, C.c_void_p(s) if s is not None else None), "net_set_side_stream")
        self.overlap = bool(on)
        self._drop_graphs()         # a captured graph froze the old fork / join plan

    # ---------------- parameters ----------------
    def view(self, spec: TensorSpec, buf: Optional[torch.Tensor] = None) -> torch.Tensor:
        base = buf if buf is not None else (self.params if spec.trainable else self.state)
        L = self.latent_dim
        if spec.part == 0:
            n = int(np.prod(spec.shape))
            return base[spec.offset:spec.offset + n].view(*spec.shape)
        lo = 0 if spec.part == 1 else L
        if len(spec.shape) == 2:      # head kernel: columns [lo, lo+L) of the [100][2L] matrix
            return base[spec.offset:spec.offset + 100 * 2 * L].view(100, 2 * L)[:, lo:lo + L]
        return base[spec.offset + lo:spec.offset + lo + L]

    def named_views(self, buf: Optional[torch.Tensor] = None, trainable_only=False):
        for s in self.specs:
            if trainable_only and not s.trainable:
                continue
            if buf is not None and not s.trainable:
                continue
            yield s.name, self.view(s, buf)

    def reset_bn_state(self):
        for s in self.specs:
            if not s.trainable:
                self.view(s).fill_(1.0 if s.name.endswith(".var") else 0.0)

    def init_weights(self, seed: Optional[int] = None):
        """Keras initialisers of the reference: he_uniform on conv kernels (models.py:217-301),
        glorot_uniform on Dense kernels, zero biases, BN gamma=1 beta=0 mean=0 var=1."""
        rng = np.random.default_rng(seed)
        for s in self.specs:
            leaf = s.name.rsplit(".", 1)[1]
            if leaf == "kernel":
                if len(s.shape) == 4:
                    lim = np.sqrt(6.0 / (s.shape[0] * s.shape[1] * s.shape[2]))
                else:
                    lim = np.sqrt(6.0 / (s.shape[0] + s.shape[1]))
                v = rng.uniform(-lim, lim, size=s.shape)
            elif leaf in ("gamma", "var"):
                v = np.ones(s.shape)
            else:
                v = np.zeros(s.shape)
            self.view(s).copy_(torch.from_numpy(v.astype(np.float32)))
        self.adam_m.zero_(); self.adam_v.zero_(); self.adam_step_count.zero_()

    def set_weights_dict(self, d: Dict[str, np.ndarray]):
        for s in self.specs:
            if s.name in d:
                self.view(s).copy_(torch.from_numpy(np.ascontiguousarray(np.asarray(d[s.name], dtype=np.float32))))

    def get_weights_dict(self, buf: Optional[torch.Tensor] = None) -> Dict[str, np.ndarray]:
        return {n: v.detach().cpu().numpy().copy() for n, v in self.named_views(buf)}

    def get_weights(self) -> List[np.ndarray]:
        """keras Model.get_weights() order ([KERAS-SEM], parity unpinned): encoder trainable,
        encoder moving statistics, decoder trainable, decoder moving statistics."""
        return [self.view(s).detach().cpu().numpy().copy() for s in self.specs]

    def _weight_orders(self):
        """The get_weights() order, and the flat "all trainable, then all moving statistics" order this
        package wrote before round 2 (lists saved then still load)."""
        keras = list(self.specs)
        flat = [s for s in self.specs if s.trainable] + [s for s in self.specs if not s.trainable]
        return [keras, flat]

    def set_weights(self, ws: Sequence[np.ndarray]):
        if len(ws) != len(self.specs):
            raise ValueError(f"expected {len(self.specs)} arrays, got {len(ws)}")
        arrs = [np.asarray(w, dtype=np.float32) for w in ws]
        shapes = [tuple(a.shape) for a in arrs]
        for order in self._weight_orders():      # accepted by the sequence of shapes
            if shapes == [tuple(s.shape) for s in order]:
                for s, w in zip(order, arrs):
                    self.view(s).copy_(torch.from_numpy(np.ascontiguousarray(w)))
                return
        for s, shp in zip(self.specs, shapes):
            if shp != tuple(s.shape):
                raise ValueError(f"{s.name}: shape {shp} != {s.shape}")

    def set_bucket_event(self, event: Optional["torch.cuda.Event"]):
        """Data parallelism in two buckets (veex_net_set_bucket_event): train_fwd_bwd records ``event`` where the
        decoder's kernel / bias gradients -- ``grad_bucket()``, one contiguous range of ``grads`` -- are final."""
        if event is not None:
            event.record()                      # materialises the HIP event (torch creates it lazily)
        self._bucket_event = event
        check(lib.veex_net_set_bucket_event(self._h, C.c_void_p(event.cuda_event) if event is not None else None),
              "net_set_bucket_event")
        self._drop_graphs()

    def grad_bucket(self):
        """(first, count) of the early gradient bucket inside ``grads``."""
        first, count = C.c_longlong(), C.c_longlong()
        check(lib.veex_net_grad_bucket(self._h, C.byref(first), C.byref(count)), "net_grad_bucket")
        return first.value, count.value

    def set_batch_divisor(self, divisor: float = 0.0):
        """Divisor of the batch-mean loss and its gradients; 0 = the batch of the call.  A rank that holds
        ``n_local`` of ``n_global`` samples sets ``n_global / world``: SUM all-reduce + Adam's 1/world is
        then the mean over the global batch whatever the shard sizes (models.py:156-163)."""
        check(lib.veex_net_set_batch_divisor(self._h, float(divisor)), "net_set_batch_divisor")

    # ---------------- workspace ----------------
    def _drop_graphs(self):
        """Forget every captured graph (forward_graphed, train_step_graphed): they hold raw pointers into the
        workspaces and the compute mode / stream plan they were captured with."""
        self.__dict__.pop("_fwd_graphs", None)
        self.__dict__.pop("_train_graphs", None)

    def _drop_workspaces(self):
        self._ws.clear()
        self._drop_graphs()

    def workspace(self, batch: int) -> torch.Tensor:
        ws = self._ws.get(batch)
        if ws is None:
            n = lib.veex_net_workspace_floats(self._h, batch)
            if n < 0:
                raise _lib.VeexError(lib.veex_last_error().decode())
            # every region is written by a kernel before a kernel reads it: no zero fill (torch.zeros launched a
            # FillFunctor over ~1 GB per new batch size)
            ws = torch.empty(n, dtype=torch.float32, device=self.device)
            if len(self._ws) > 4:
                # evict the workspaces no captured graph points into (a graph entry keeps its own reference, so its
                # memory stays valid either way; this only bounds what the cache itself holds)
                pinned = {k[0] for k in self.__dict__.get("_fwd_graphs", {})} | set(self.__dict__.get("_train_graphs", {}))
                for b in [b for b in self._ws if b not in pinned]:
                    del self._ws[b]
            self._ws[batch] = ws
        return ws

    def ws_region(self, batch: int, name: str) -> torch.Tensor:
        n = C.c_longlong()
        off = lib.veex_net_ws_offset(self._h, batch, name.encode(), C.byref(n))
        if off < 0:
            raise KeyError(name)
        return self.workspace(batch)[off:off + n.value]

    # ---------------- calls ----------------
    def _eps_args(self, eps, batch):
        if self.variant != "vae":
            return None, 0
        if eps is None:
            return None, _lib.F_PHILOX
        e = _as_device_f32(eps, self.device)
        if tuple(e.shape) != (batch, self.latent_dim):
            raise ValueError(f"epsilon shape {tuple(e.shape)} != {(batch, self.latent_dim)}")
        return e, 0

    def _x(self, x) -> torch.Tensor:
        t = _as_device_f32(x, self.device)
        if t.dim() == 4 and t.shape[1:] == (40, 40, 1):
            pass
        elif t.dim() == 3 and t.shape[1:] == (40, 40):
            pass
        else:
            raise ValueError(f"expected (B,40,40,1) input, got {tuple(t.shape)}")
        return t

    def encode(self, x, eps=None):
        """Encoder.call: returns (z_mean, z_log_var, z) device tensors; AE: (z, None, z)."""
        x = self._x(x)
        B, L = x.shape[0], self.latent_dim
        ws = self.workspace(B)
        e, flags = self._eps_args(eps, B)
        first = self.samples_seen
        if flags & _lib.F_PHILOX:
            self.samples_seen += B
        vae = self.variant == "vae"
        if self.via_ops:
            zm, zlv, z = torch.ops.veex.encode(self._h, self.params, self.state, x, e, flags, self.philox_seed, first,
                                               ws, L, vae)
        else:
            z = torch.empty(B, L, dtype=torch.float32, device=self.device)
            zm, zlv = (torch.empty_like(z), torch.empty_like(z)) if vae else (None, None)
            check(lib.veex_net_encode(self._h, ptr(self.params), ptr(self.state), ptr(x), ptr(e), B, flags,
                                      self.philox_seed, first, ptr(ws), ws.numel(), ptr(zm), ptr(zlv), ptr(z),
                                      _lib.current_stream()), "net_encode")
        return (zm, zlv, z) if vae else (z, None, z)

    def decode(self, z, want_logits=False):
        z = _as_device_f32(z, self.device)
        if z.dim() != 2 or z.shape[1] != self.latent_dim:
            raise ValueError(f"expected (B,{self.latent_dim}) latent, got {tuple(z.shape)}")
        B = z.shape[0]
        ws = self.workspace(B)
        if self.via_ops:
            recon, logits = torch.ops.veex.decode(self._h, self.params, self.state, z, ws, bool(want_logits))
        else:
            recon = torch.empty(B, 40, 40, 1, dtype=torch.float32, device=self.device)
            logits = torch.empty_like(recon) if want_logits else None
            check(lib.veex_net_decode(self._h, ptr(self.params), ptr(self.state), ptr(z), B, ptr(ws), ws.numel(),
                                      ptr(recon), ptr(logits), _lib.current_stream()), "net_decode")
        return (recon, logits) if want_logits else recon

    def forward(self, x, eps=None, want_recon=True, accumulate_metrics=False, zmean_decode=False,
                first_sample: Optional[int] = None):
        """VAE.call + losses (test_step): returns (recon or None, losses[3] device tensor)."""
        x = self._x(x)
        B = x.shape[0]
        ws = self.workspace(B)
        e, flags = self._eps_args(eps, B)
        if zmean_decode:
            flags |= _lib.F_ZMEAN_DECODE
        first = first_sample
        if first is None:
            first = self.samples_seen
            if flags & _lib.F_PHILOX:
                self.samples_seen += B
        metrics = self.metrics if accumulate_metrics else None
        if self.via_ops:
            recon, losses = torch.ops.veex.forward(self._h, self.params, self.state, x, e, flags, self.philox_seed,
                                                   first, ws, metrics, bool(want_recon))
            return (recon if want_recon else None), losses
        recon = torch.empty(B, 40, 40, 1, dtype=torch.float32, device=self.device) if want_recon else None
        losses = torch.empty(3, dtype=torch.float32, device=self.device)
        check(lib.veex_net_forward(self._h, ptr(self.params), ptr(self.state), ptr(x), ptr(e), B, flags,
                                   self.philox_seed, first, ptr(ws), ws.numel(), ptr(recon), ptr(losses),
                                   ptr(metrics), _lib.current_stream()), "net_forward")
        return recon, losses

    def forward_graphed(self, x, eps=None, zmean_decode=False):
        """``forward`` replayed from a captured HIP graph: serving at a fixed small batch, where a forward costs the
        host more to enqueue (~18 launches through Python + ctypes) than the GPU to run.  The first call per
        (batch, epsilon given?, zmean_decode) captures ``forward`` on static input buffers; later calls copy ``x``
        (and ``eps``) into them and replay.  The library only enqueues on the caller's stream and creates nothing, so
        the whole call captures.  The VAE's sampled path needs ``eps`` (an injected epsilon, or ``zmean_decode=True``
        for the deterministic reconstruction): a Philox seed / counter would be frozen into the graph.  Returns
        (recon, losses) in STATIC tensors that the next call on the same key overwrites."""
        x = self._x(x)
        B = x.shape[0]
        if self.variant == "vae" and eps is None and not zmean_decode:
            raise ValueError("forward_graphed: pass eps (or zmean_decode=True); a Philox counter cannot live in a graph")
        e = _as_device_f32(eps, self.device) if (eps is not None and self.variant == "vae") else None
        key = (B, e is not None, bool(zmean_decode))
        graphs = self.__dict__.setdefault("_fwd_graphs", {})
        ent = graphs.get(key)
        if ent is None:
            sx = torch.zeros(B, 40, 40, 1, dtype=torch.float32, device=self.device)
            se = torch.zeros(B, self.latent_dim, dtype=torch.float32, device=self.device) if e is not None else None
            dummy = se if se is not None else (torch.zeros(B, self.latent_dim, device=self.device) if self.variant == "vae" else None)
            self.workspace(B)                                  # allocated outside the capture
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(side):                      # warm-up off the default stream, as capture requires
                self.forward(sx, dummy, zmean_decode=zmean_decode, first_sample=0)
            torch.cuda.current_stream(self.device).wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                out = self.forward(sx, dummy, zmean_decode=zmean_decode, first_sample=0)
            # the entry owns everything the graph's kernels point at: the workspace of this batch size and the
            # zero epsilon of the zmean_decode key (both would otherwise go back to the caching allocator)
            ent = graphs[key] = (g, sx, se, out, self.workspace(B), dummy)
        g, sx, se, out = ent[:4]
        sx.copy_(x.reshape(sx.shape))
        if se is not None:
            se.copy_(e)
        g.replay()
        return out

    # ---------------- captured train step (small batches) ----------------
    GRAPH_MIN_BATCH, GRAPH_MAX_BATCH = 192, 384     # VAE.fit's automatic window (measured, profiles/r03_fit_batches.jsonl)

    def train_step_graphed(self, x_rows: torch.Tensor, index: torch.Tensor, lr: float):
        """One whole optimisation step -- gather of the batch, veex_net_train_fwd_bwd, veex_adam -- replayed from a
        HIP graph: at the reference's batch sizes (32-128, train.py:104-109) the ~60 launches of a step cost the host
        more to enqueue through Python than the GPU needs to run them.  ``x_rows`` is the resident set, ``index``
        (device int64, length B) the rows of this batch.  Captured once per (B, lr, set); the Philox epsilon of a
        replay is fresh because the kernel reads the sample counter from a device word (veex_net_set_sample_counter)
        that the graph itself advances by B; ``samples_seen`` on the host is advanced alongside, and the two are
        re-aligned whenever an eager call moved the host count.  Same kernels, same order, same arithmetic as the
        eager step: bit-identical results (tests/test_gpu_driver.py).  Single process only (no all-reduce inside)."""
        B = int(index.shape[0])
        graphs = self.__dict__.setdefault("_train_graphs", {})
        ent = graphs.get(B)
        key = (float(lr), x_rows.data_ptr(), tuple(x_rows.shape))
        if ent is not None and ent["key"] != key:
            ent = None
        if ent is None:
            ctr = torch.zeros(1, dtype=torch.int64, device=self.device)
            sidx = torch.zeros(B, dtype=torch.int64, device=self.device)
            sx = torch.empty(B, 40, 40, 1, dtype=torch.float32, device=self.device)
            ws = self.workspace(B)
            rows2d = x_rows.view(x_rows.shape[0], -1)

            def body():
                torch.index_select(rows2d, 0, sidx, out=sx.view(B, -1))
                self.train_fwd_bwd(sx, None, accumulate_metrics=True, first_sample=0)
                check(lib.veex_counter_add(C.c_void_p(ctr.data_ptr()), B, _lib.current_stream()), "counter_add")
                self.adam_step(lr)

            # warm-up off the default stream (function attributes, allocator) on a snapshot of everything a step
            # changes, restored afterwards: the capture run itself executes nothing
            snap = [t.clone() for t in (self.params, self.adam_m, self.adam_v, self.adam_step_count, self.metrics)]
            check(lib.veex_net_set_sample_counter(self._h, C.c_void_p(ctr.data_ptr())), "set_sample_counter")
            try:
                side = torch.cuda.Stream(device=self.device)
                side.wait_stream(torch.cuda.current_stream(self.device))
                with torch.cuda.stream(side):
                    body()
                torch.cuda.current_stream(self.device).wait_stream(side)
                for t, s in zip((self.params, self.adam_m, self.adam_v, self.adam_step_count, self.metrics), snap):
                    t.copy_(s)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    body()
            finally:
                check(lib.veex_net_set_sample_counter(self._h, None), "set_sample_counter")
            ent = graphs[B] = {"g": g, "ctr": ctr, "idx": sidx, "x": sx, "ws": ws, "key": key, "rows": x_rows, "dev_count": None}
        if ent["dev_count"] != self.samples_seen:          # an eager call (validation, a ragged batch) moved the count
            ent["ctr"].fill_(self.samples_seen)
        ent["idx"].copy_(index)
        ent["g"].replay()
        self.samples_seen += B
        ent["dev_count"] = self.samples_seen
        return self.grads[self.n_params:self.n_params + 3]

    def train_fwd_bwd(self, x, eps=None, accumulate_metrics=True, first_sample: Optional[int] = None):
        """Forward + loss + every gradient (VAE.train_step up to tape.gradient).  The three loss
        scalars land in grads[-4:-1]."""
        x = self._x(x)
        B = x.shape[0]
        ws = self.workspace(B)
        e, flags = self._eps_args(eps, B)
        if first_sample is None:
            first_sample = self.samples_seen
            if flags & _lib.F_PHILOX:
                self.samples_seen += B
        losses = self.grads[self.n_params:self.n_params + 3]
        metrics = self.metrics if accumulate_metrics else None
        if self.via_ops:
            torch.ops.veex.train_fwd_bwd(self._h, self.params, self.state, x, e, flags, self.philox_seed,
                                         first_sample, ws, self.grads, metrics)
            return losses
        check(lib.veex_net_train_fwd_bwd(self._h, ptr(self.params), ptr(self.state), ptr(x), ptr(e), B, flags,
                                         self.philox_seed, first_sample, ptr(ws), ws.numel(), ptr(self.grads),
                                         ptr(losses), ptr(metrics), _lib.current_stream()), "net_train_fwd_bwd")
        return losses

    def adam_step(self, lr: float, grad_scale: float = 1.0):
        """optimizer.apply_gradients (models.py:169) with keras legacy Adam."""
        if self.via_ops:
            torch.ops.veex.adam(self.params, self.grads[:self.n_params], self.adam_m, self.adam_v, self.adam_step_count,
                                self.adam_alpha, float(lr), float(grad_scale), ADAM_B1, ADAM_B2, ADAM_EPS)
            return
        check(lib.veex_adam(ptr(self.params), ptr(self.grads), ptr(self.adam_m), ptr(self.adam_v), self.n_params,
                            ptr(self.adam_step_count), ptr(self.adam_alpha), float(lr), float(grad_scale),
                            ADAM_B1, ADAM_B2, ADAM_EPS, _lib.current_stream()), "adam")

    def attribute(self, x, units: Iterable[int], grad_x_input: bool = False) -> torch.Tensor:
        """Saliency maps d z_mean[:, j] / d x, shape (len(units), B, 40, 40)."""
        x = self._x(x)
        B = x.shape[0]
        units = [int(u) for u in units]
        ws = self.workspace(B)
        if self.via_ops:
            return torch.ops.veex.attribute(self._h, self.params, self.state, x, units, bool(grad_x_input), ws)
        out = torch.empty(len(units), B, 40, 40, dtype=torch.float32, device=self.device)
        arr = (C.c_int * len(units))(*units)
        check(lib.veex_net_attribute(self._h, ptr(self.params), ptr(self.state), ptr(x), B, arr, len(units),
                                     1 if grad_x_input else 0, ptr(ws), ws.numel(), ptr(out),
                                     _lib.current_stream()), "net_attribute")
        return out
