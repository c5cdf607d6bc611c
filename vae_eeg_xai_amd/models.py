"""Drop-in counterpart of the reference's ``models.py`` (sample, Encoder, Decoder, VAE) on the
MI355X HIP library.  Same class names, constructor arguments, call signatures and return
structure (reference models.py:10-22,122-312); everything numerical runs in libveex.so.

Differences a caller can observe (documented, not silent):
  * returned tensors are ``DeviceArray`` objects (ndarray protocol; data stays in HBM until used);
  * ``sample`` / ``Encoder.__call__`` accept an optional injected ``epsilon``; without it epsilon
    comes from the device Philox generator (the reference's tf.random.normal is unseeded);
  * initial weights are drawn from NumPy's generator (same distributions as he_uniform /
    glorot_uniform, different stream than TensorFlow's);
  * ``save_weights`` writes ``<path>.npz`` (a TF object-graph checkpoint cannot be produced).
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from .net import Net, _as_device_f32
from .keras_compat import Adam, EarlyStopping, History, tf  # noqa: F401  (re-exported like the reference)
from . import dist as vdist

keras = tf.keras


class DeviceArray(np.lib.mixins.NDArrayOperatorsMixin):
    """ndarray-protocol view of a device tensor: np.asarray / indexing / .shape / .reshape /
    np.quantile / np.copy / arithmetic work as on the tensors the reference returns
    (analysis.py:38-52,243,492).  The data stays in HBM until the host looks at it; the first host-side look
    (np.asarray, a scalar index, arithmetic) copies it ONCE and later looks reuse that copy -- analysis.py:255-264,
    323-333 index ``z_mean_blink[i, j]`` in O(N * L * quantiles) Python loops, which must not be one device
    round trip each.  The wrapped tensor is treated as immutable (the library never writes into a returned tensor)."""

    __array_priority__ = 100

    def __init__(self, t: torch.Tensor):
        self._t = t
        self._host = None

    # ndarray protocol
    def _np(self):
        if self._host is None:
            self._host = self._t.detach().cpu().numpy()
        return self._host

    def __array__(self, dtype=None, copy=None):
        a = self._np()
        if dtype is not None and np.dtype(dtype) != a.dtype:
            return a.astype(dtype)
        return a.copy() if copy else a

    def __array_ufunc__(self, ufunc, method, *inputs, out=None, **kwargs):
        # z - 1, z * mask, np.exp(z), z > q ...: NumPy on the cached host copy, like the ndarray-behaving tensors
        # of the reference (np_config.enable_numpy_behavior, analysis.py:23)
        if out is not None:
            return NotImplemented
        args = [i._np() if isinstance(i, DeviceArray) else i for i in inputs]
        return getattr(ufunc, method)(*args, **kwargs)

    def numpy(self):
        return self._np()

    @property
    def shape(self):
        return tuple(self._t.shape)

    @property
    def dtype(self):
        return np.dtype("float32")

    @property
    def ndim(self):
        return self._t.dim()

    @property
    def size(self):
        return int(self._t.numel())

    @property
    def T(self):
        return self._np().T

    def __len__(self):
        return self._t.shape[0]

    def __getitem__(self, idx):
        # Scalars, slices and anything else once a host copy exists: plain NumPy on that copy (NumPy's own
        # IndexError for an out-of-range index).  A fancy index (index array / mask) on a set that is still only in
        # HBM is applied there and only the selection moves (analysis.py:46,581 pick rows of whole-set outputs) --
        # after a host-side bounds check: an out-of-range index inside the device gather is a device-side assert
        # that poisons the HIP context.
        if self._host is None and _is_fancy(idx):
            try:
                tidx = _device_index(idx, self._t.device, self._t.shape)
            except (TypeError, ValueError):
                return self._np()[idx]            # index forms torch lacks
            a = self._t[tidx].detach().cpu().numpy()
            return a[()] if a.ndim == 0 else a
        return self._np()[idx]

    def __iter__(self):
        return iter(self._np())

    def __float__(self):
        return float(self._np())

    def reshape(self, *shape):
        return self._np().reshape(*shape)

    def astype(self, dtype):
        return self._np().astype(dtype)

    def copy(self):
        return self._np().copy()

    def min(self, *a, **k): return self._np().min(*a, **k)
    def max(self, *a, **k): return self._np().max(*a, **k)
    def mean(self, *a, **k): return self._np().mean(*a, **k)
    def sum(self, *a, **k): return self._np().sum(*a, **k)
    def std(self, *a, **k): return self._np().std(*a, **k)

    def __repr__(self):
        return f"DeviceArray(shape={self.shape}, device={self._t.device})"


def _is_fancy(idx):
    parts = idx if isinstance(idx, tuple) else (idx,)
    return any(isinstance(i, (list, np.ndarray, torch.Tensor, DeviceArray)) for i in parts)


def _device_index(idx, device, shape, axis=0):
    """NumPy-style index -> torch index with array parts bounds-checked on the host (IndexError like NumPy) and
    moved to the device."""
    if isinstance(idx, tuple):
        if any(i is None or i is Ellipsis for i in idx):
            raise TypeError("newaxis / Ellipsis: host path")
        return tuple(_device_index(i, device, shape, k) for k, i in enumerate(idx))
    if isinstance(idx, DeviceArray):
        idx = idx._np()
    if isinstance(idx, torch.Tensor):
        idx = idx.detach().cpu().numpy()
    if isinstance(idx, (list, np.ndarray)):
        a = np.asarray(idx)
        n = int(shape[axis]) if axis < len(shape) else 0
        if a.dtype == np.bool_:
            if a.ndim != 1 or a.shape[0] != n:
                raise TypeError("multi-dimensional or mis-sized mask: host path (NumPy raises what it raises)")
        else:
            if a.size and not np.issubdtype(a.dtype, np.integer):
                raise IndexError("arrays used as indices must be of integer (or boolean) type")
            a = a.astype(np.int64)
            if a.size and (a.min() < -n or a.max() >= n):
                raise IndexError(f"index {int(a.max() if a.max() >= n else a.min())} is out of bounds for axis {axis} with size {n}")
        return torch.from_numpy(np.ascontiguousarray(a)).to(device)
    if isinstance(idx, np.integer):
        return int(idx)
    return idx


def _host_rng_eps(shape, generator):
    if generator is None:
        return None
    if isinstance(generator, np.random.Generator):
        return generator.standard_normal(shape).astype(np.float32)
    raise TypeError("generator must be a numpy.random.Generator")


def sample(z_mean, z_log_var, epsilon=None, generator=None):
    """Reparameterisation trick (reference models.py:10-22): z = mean + exp(0.5*log_var) * eps.
    Runs the fused HIP reparameterisation kernel; ``epsilon`` (B,L) injects the noise.  With torch tensors
    that require grad the result is differentiable (torch.ops.veex.sample has a registered backward)."""
    from . import _lib
    from ._lib import lib, check, ptr
    dev = torch.device("cuda", torch.cuda.current_device())
    keep_graph = any(isinstance(t, torch.Tensor) and t.requires_grad for t in (z_mean, z_log_var))
    # differentiable path: .float() here, in Python, so that autograd records the cast (a float64 / half input then
    # gets its gradient back in its own dtype); the op itself only sees float32
    zm = z_mean.to(dev).float() if keep_graph else _as_device_f32(z_mean, dev)
    zlv = z_log_var.to(dev).float() if keep_graph else _as_device_f32(z_log_var, dev)
    B, L = zm.shape
    if epsilon is None:
        epsilon = _host_rng_eps((B, L), generator)
    if epsilon is None:      # unseeded like the reference's tf.random.normal: Philox with a fresh seed
        eps = torch.empty(B, L, dtype=torch.float32, device=dev)
        seed = int(np.random.SeedSequence().entropy % (1 << 63))
        mulv = torch.cat([zm.detach(), zlv.detach()], dim=1).contiguous()
        scratch = torch.empty_like(eps)
        check(lib.veex_reparam_kl_fwd(ptr(mulv), ptr(eps), ptr(scratch), None, B, L, 1, seed, 0, _lib.current_stream()))
    else:
        eps = _as_device_f32(epsilon, dev)
    z = torch.ops.veex.sample(zm, zlv, eps)
    return z if keep_graph else DeviceArray(z)


class _LayerHandle:
    """Named handle on one layer's tensors (the reference exposes conv1.., dense, z_mean, ... attributes)."""

    def __init__(self, owner, prefix):
        self._owner, self._prefix = owner, prefix

    def get_weights(self):
        net = self._owner._net
        return [net.view(s).detach().cpu().numpy().copy() for s in net.specs if s.name.startswith(self._prefix + ".")]


class Encoder:
    """Encoder(latent_dimension) -> (z_mean, z_log_var, z)  (reference models.py:211-253)."""

    _variant = "vae"

    def __init__(self, latent_dimension):
        self.latent_dim = int(latent_dimension)
        self._net = Net(self.latent_dim, self._variant)
        self._net.init_weights(seed=42)
        for nm in ("conv1", "conv2", "conv3", "dense"):
            setattr(self, nm, _LayerHandle(self, f"enc.{nm}"))
        self.flatten = None
        if self._variant == "vae":
            self.z_mean = _LayerHandle(self, "enc.z_mean")
            self.z_log_var = _LayerHandle(self, "enc.z_log_var")
            self.sampling = sample

    def __call__(self, inputs, training=None, mask=None, epsilon=None):
        # ``training`` is accepted and ignored exactly like the reference: BatchNorm always runs in
        # inference mode there (models.py:244-253 never forwards it)
        zm, zlv, z = self._net.encode(inputs, epsilon)
        return DeviceArray(zm), DeviceArray(zlv), DeviceArray(z)

    call = __call__


class Decoder:
    """Decoder() : z -> (B,40,40,1) reconstruction  (reference models.py:256-312)."""

    _variant = "vae"

    def __init__(self):
        self._net: Optional[Net] = None
        for nm in ("dense", "deconv1", "deconv2", "deconv3", "deconv4", "deconv5", "deconv6"):
            setattr(self, nm, _LayerHandle(self, f"dec.{nm}"))

    def _bind(self, net: Net):
        self._net = net

    def __call__(self, inputs, training=None, mask=None):
        z = inputs._t if isinstance(inputs, DeviceArray) else inputs
        if self._net is None:      # stand-alone decoder: latent size from the input
            L = int(np.asarray(z.shape)[1])
            self._net = Net(L, self._variant)
            self._net.init_weights(seed=42)
        return DeviceArray(self._net.decode(z))

    call = __call__


class _Mean:
    """keras.metrics.Mean look-alike backed by the device-side accumulators."""

    def __init__(self, name, owner, idx):
        self.name, self._owner, self._idx = name, owner, idx

    def result(self):
        m = self._owner._net.metrics
        return float((m[self._idx] / torch.clamp(m[3], min=1.0)).item())

    def reset_state(self):
        self._owner._net.metrics.zero_()


class VAE:
    """VAE(encoder, decoder, epochs, l_rate, batch_size, patience)  (reference models.py:122-207)."""

    _metric_names = ("loss", "reconstruction_loss", "kl_loss")
    _variant = "vae"

    def __init__(self, encoder, decoder, epochs=None, l_rate=None, batch_size=None, patience=None, **kwargs):
        self.encoder = encoder
        self.decoder = decoder
        self.epochs = epochs  # For grid search (kept like the reference; never read)
        self.l_rate = l_rate
        self.batch_size = batch_size
        self.patience = patience
        self._net: Net = encoder._net
        if decoder._net is not None and decoder._net is not self._net:
            # adopt the decoder's existing weights
            src = decoder._net.get_weights_dict()
            self._net.set_weights_dict({k: v for k, v in src.items() if k.startswith("dec.")})
        decoder._bind(self._net)
        self.optimizer = None
        self.stop_training = False
        self.grad_buckets = int(kwargs.pop("grad_buckets", 2))   # data parallelism: 2 = overlapped exchange, 1 = one all-reduce
        # the two Keras readings nothing in the reference pins (Net.set_semantics): {"bn": "inference"|"batch",
        # "bce": "logits"|"probs"}.  bn="batch" under data parallelism normalises each rank's slice on its own
        # statistics, as Keras's per-replica BatchNormalization does.
        semantics = kwargs.pop("semantics", None)
        if semantics:
            self._net.set_semantics(**semantics)
        # engine of the large GEMMs (Net.set_compute): "f32" (fp32 MFMA, default), "f32x3" (fp32 results on the bf16 matrix
        # cores, same parity bar, 8 % faster at batch 512), "bf16" (own tolerance)
        compute = kwargs.pop("compute", None)
        if compute:
            self._net.set_compute(compute)
        self.total_loss_tracker = _Mean("total_loss", self, 0)
        self.reconstruction_loss_tracker = _Mean("reconstruction_loss", self, 1)
        self.kl_loss_tracker = _Mean("kl_loss", self, 2)

    # ---- keras.Model surface used by the reference ----
    @property
    def metrics(self):
        return [self.total_loss_tracker, self.reconstruction_loss_tracker, self.kl_loss_tracker]

    @property
    def trainable_weights(self):
        return [v for _, v in self._net.named_views(trainable_only=True)]

    def compile(self, optimizer=None, **kwargs):
        self.optimizer = optimizer if optimizer is not None else Adam()

    def _lr(self):
        if self.optimizer is None:
            raise RuntimeError("call compile(optimizer) before training")
        return float(self.optimizer.learning_rate)

    def _logs(self, prefix=""):
        m = self._net.metrics.detach().cpu().numpy()
        cnt = max(float(m[3]), 1.0)
        vals = (m[0] / cnt, m[1] / cnt, m[2] / cnt)
        return {prefix + k: float(v) for k, v in zip(self._metric_names, vals)}

    def call(self, inputs, training=None, mask=None, epsilon=None):
        """decoder(encoder(x)[2]) -- decodes the *sampled* z like the reference (models.py:135-138)."""
        recon, _ = self._net.forward(inputs, epsilon, want_recon=True)
        return DeviceArray(recon)

    __call__ = call

    def train_step(self, data, epsilon=None):
        """One optimisation step (reference models.py:148-180); returns the running means.  Under
        torch.distributed every rank passes the same (global) batch and works on its own slice of it."""
        x, _labels = data
        self._step(x, epsilon)
        return self._logs()

    def test_step(self, data, epsilon=None):
        x, _labels = data
        self._net.forward(x, epsilon, want_recon=False, accumulate_metrics=True)
        return self._logs()

    def _step(self, x, epsilon=None, index=None):
        """One optimisation step on a GLOBAL batch: ``x`` itself, or rows ``index`` (device int64) of the
        resident set ``x``.  Data parallelism: rank r takes the contiguous slice
        ``shard_range(n_global, r, world)``; the Philox epsilon of a sample is keyed by its global index
        (``samples_seen`` counts GLOBAL samples on every rank), and the loss is divided by
        ``n_global / world`` so that ONE sum all-reduce of the flat gradient buffer followed by Adam's
        1/world is the mean over the global batch (models.py:156-163) for any shard sizes, empty ones
        included.  With one process this is the plain step."""
        net = self._net
        world, rank = vdist.world_size(), vdist.rank()
        red = self._reducer(world)
        ran = self._shard_fwd_bwd(x, epsilon, index, world, rank)
        if world > 1:
            if red is not None:
                red.reduce_(ran)                       # decoder bucket overlapped with the encoder's backward pass
            else:
                vdist.allreduce_sum_(net.grads)        # gradients and the 3 loss scalars in one collective
            net.metrics[:3] += net.grads[net.n_params:net.n_params + 3] / world
            net.metrics[3] += 1.0
        net.adam_step(self._lr(), 1.0 / world)

    def _reducer(self, world):
        """The two-bucket gradient exchange (dist.TwoBucketReducer) of a data-parallel run, or None: one process, or
        ``grad_buckets=1`` / VEEX_GRAD_BUCKETS=1 (the single all-reduce, kept as the tested reference)."""
        import os
        want = int(os.environ.get("VEEX_GRAD_BUCKETS", getattr(self, "grad_buckets", 2)))
        if world == 1 or want < 2:
            if getattr(self, "_red", None) is not None:
                self._red.close()
                self._red = None
            return None
        if getattr(self, "_red", None) is None:
            self._red = vdist.TwoBucketReducer(self._net)
        return self._red

    def _shard_fwd_bwd(self, x, epsilon, index, world, rank):
        """Rank ``rank``'s share of one global step: its slice of the batch through
        veex_net_train_fwd_bwd, gradients (already divided for the GLOBAL mean) into ``net.grads``.  Returns
        whether a backward pass ran (False: an empty shard, the buffer was zeroed)."""
        net = self._net
        n_global = int(index.shape[0]) if index is not None else int(x.shape[0])
        lo, hi = vdist.shard_range(n_global, rank, world)
        first = net.samples_seen + lo
        net.samples_seen += n_global
        if world == 1:
            xb = x if index is None else x.index_select(0, index)
            net.train_fwd_bwd(xb, epsilon, accumulate_metrics=True, first_sample=first)
            return True
        if hi == lo:
            net.grads.zero_()                          # more ranks than samples: contribute nothing, still reduce
            return False
        xb = x[lo:hi] if index is None else x.index_select(0, index[lo:hi])
        eb = None if epsilon is None else epsilon[lo:hi]
        net.set_batch_divisor(n_global / world)
        try:
            net.train_fwd_bwd(xb, eb, accumulate_metrics=False, first_sample=first)
        finally:
            net.set_batch_divisor(0.0)
        return True

    def _validate(self, xv, batch_size):
        """test_step over the validation set (models.py:182-207), sharded like the training batches; the
        four metric accumulators are summed over ranks once at the end, so every rank logs the same values
        (and EarlyStopping stops all ranks in the same epoch)."""
        net = self._net
        world, rank = vdist.world_size(), vdist.rank()
        net.metrics.zero_()
        for s in range(0, xv.shape[0], batch_size):
            n_global = min(batch_size, xv.shape[0] - s)
            lo, hi = vdist.shard_range(n_global, rank, world)
            first = net.samples_seen + lo
            net.samples_seen += n_global
            if world == 1:
                net.forward(xv[s:s + n_global], None, want_recon=False, accumulate_metrics=True, first_sample=first)
            elif hi > lo:
                net.set_batch_divisor(n_global / world)
                try:
                    net.forward(xv[s + lo:s + hi], None, want_recon=False, accumulate_metrics=True, first_sample=first)
                finally:
                    net.set_batch_divisor(0.0)
            else:
                net.metrics[3] += 1.0
        if world > 1:
            vdist.allreduce_sum_(net.metrics)
            net.metrics /= world

    def train_on_batch(self, x, y=None):
        self._net.metrics.zero_()
        self._step(_as_device_f32(x, self._net.device))
        logs = self._logs()
        return [logs[k] for k in self._metric_names]

    def fit(self, x=None, y=None, batch_size=None, epochs=1, verbose=1, callbacks=None, validation_data=None,
            shuffle=True, seed=None, **kwargs):
        """keras Model.fit as the reference calls it (train.py:141-142,196-197): positional
        (x, y, batch_size, epochs); y is ignored (train_step unpacks and drops it); shuffles every
        epoch, keeps the last partial batch, logs the unweighted running mean over batches.
        Under torch.distributed ``batch_size`` is the GLOBAL batch: all ranks walk the same permutation
        (rank 0's seed is broadcast when ``seed`` is None) and each takes its slice of every batch, the
        ragged last one included (see _step)."""
        net = self._net
        batch_size = int(batch_size or 32)
        world = vdist.world_size()
        xd = _as_device_f32(x, net.device)
        N = xd.shape[0]
        xv = None
        if validation_data is not None:
            xv = _as_device_f32(validation_data[0], net.device)
        callbacks = list(callbacks or [])
        for cb in callbacks:
            if hasattr(cb, "on_train_begin"):
                cb.on_train_begin()
        hist = History()
        if world > 1:
            seed = vdist.broadcast_int(int(np.random.SeedSequence().entropy % (1 << 62)) if seed is None else int(seed))
        rng = np.random.default_rng(seed)
        self.stop_training = False
        # Full batches can be replayed from a captured HIP graph (Net.train_step_graphed: gather + train_fwd_bwd + Adam,
        # bit-identical to the eager step).  Measured (profiles/r03_fit_batches.jsonl): at the reference's batch sizes
        # (32-128, train.py:104-109) the step is NOT bound by the host's enqueues but by the GPU-side chain of its ~60
        # dependent launches, which a graph replays one after the other just the same -- eager is 10-17 % faster there;
        # the graph wins (6 %) around batch 256, where the host starts to fall behind.  graph=True / False forces it.
        graph = kwargs.get("graph", None)
        if graph is None:
            graph = net.GRAPH_MIN_BATCH <= batch_size <= net.GRAPH_MAX_BATCH
        use_graph = bool(graph) and world == 1 and N >= batch_size and net.semantics["bn"] == "inference"
        for epoch in range(int(epochs)):
            perm = rng.permutation(N) if shuffle else np.arange(N)
            perm_d = torch.from_numpy(perm).to(net.device)
            net.metrics.zero_()
            for s in range(0, N, batch_size):
                idx = perm_d[s:s + batch_size]
                if use_graph and idx.shape[0] == batch_size:
                    net.train_step_graphed(xd, idx, self._lr())      # full batches: one graph replay per step
                else:
                    self._step(xd, None, index=idx)                  # ragged last batch, large batches, data parallel
            logs = self._logs()
            if xv is not None:
                self._validate(xv, batch_size)
                logs.update(self._logs("val_"))
            hist.append(epoch, logs)
            if verbose and vdist.rank() == 0:
                print(f"Epoch {epoch + 1}/{epochs} - " + " - ".join(f"{k}: {v:.4f}" for k, v in logs.items()))
            for cb in callbacks:
                if hasattr(cb, "on_epoch_end"):
                    cb.on_epoch_end(epoch, logs)
                if getattr(cb, "stop_training", False):
                    self.stop_training = True
            if self.stop_training:
                break
        self.history = hist
        return hist

    def predict(self, x, batch_size=32, verbose=0, epsilon=None, **kwargs):
        """keras Model.predict: batches of 32 by default, stochastic (sampled z) like VAE.call."""
        net = self._net
        xd = _as_device_f32(x, net.device)
        outs = []
        for s in range(0, xd.shape[0], int(batch_size)):
            e = None if epsilon is None else epsilon[s:s + int(batch_size)]
            recon, _ = net.forward(xd[s:s + int(batch_size)], e, want_recon=True)
            outs.append(recon)
        return torch.cat(outs, dim=0).cpu().numpy()

    def evaluate(self, x, y=None, batch_size=32, verbose=0, **kwargs):
        net = self._net
        xd = _as_device_f32(x, net.device)
        net.metrics.zero_()
        for s in range(0, xd.shape[0], int(batch_size)):
            net.forward(xd[s:s + int(batch_size)], None, want_recon=False, accumulate_metrics=True)
        logs = self._logs()
        return [logs[k] for k in self._metric_names]

    # ---- weights ----
    def get_weights(self):
        return self._net.get_weights()

    def set_weights(self, weights):
        """keras Model.set_weights order.  Also accepts the list a real Keras run pickled
        with ``vae.get_weights()`` (reference train.py:287-289): Keras may append the metric trackers'
        scalar ``total`` / ``count`` variables after the layer weights; trailing scalars are dropped."""
        weights = [np.asarray(w) for w in weights]
        n = len(self._net.specs)
        if len(weights) > n and all(w.size == 1 for w in weights[n:]):
            weights = weights[:n]
        self._net.set_weights(weights)

    def load_keras_weight_list(self, path, allow_pickle=False):
        """Weights exported from the reference: a ``.npz`` with arrays ``arr_0 .. arr_k`` in
        ``get_weights()`` order (np.savez(path, *vae.get_weights())), or -- only with
        ``allow_pickle=True``, because unpickling executes code -- the pickle of that list which the
        reference itself writes (train.py:287-289)."""
        path = str(path)
        if path.endswith(".npz"):
            with np.load(path, allow_pickle=False) as f:
                ws = [f[f"arr_{i}"] for i in range(len(f.files))]
        elif allow_pickle:
            import pickle
            with open(path, "rb") as fp:
                ws = pickle.load(fp)
        else:
            raise ValueError("a pickled weight list is only read with allow_pickle=True (it executes code); "
                             "re-export it with np.savez(path, *weights)")
        self.set_weights(ws)

    def decode_masked(self, z, unit, value="median"):
        """Latent masking of the reference's analysis (analysis.py:574-581): overwrite latent unit
        ``unit`` of every row of ``z`` with its median over the set (np.median semantics) or with
        ``value``, then decode.  Everything stays on the device."""
        from .latent_stats import column_medians
        zt = _as_device_f32(z, self._net.device).clone()
        j = int(unit)
        if not 0 <= j < zt.shape[1]:
            raise IndexError(f"latent unit {j} out of range")
        if isinstance(value, str) and value == "median":
            zt[:, j] = column_medians(zt)[j]          # radix-select kernel; (a + b) / 2 in fp32 like np.median
        else:
            zt[:, j] = float(value)
        return DeviceArray(self._net.decode(zt))

    def save_weights(self, filepath, save_format=None, **kwargs):
        d = self._net.get_weights_dict()
        np.savez(str(filepath) + ".npz", **d)

    def load_weights(self, filepath, **kwargs):
        path = str(filepath)
        if not path.endswith(".npz"):
            path += ".npz"
        with np.load(path, allow_pickle=False) as f:
            self._net.set_weights_dict({k: f[k] for k in f.files})

    # ---- gradient attribution (north_star item 3; not in the reference) ----
    def attribute(self, x, units, grad_x_input=False, batch_size=4096):
        """Saliency / gradient*input maps of latent units wrt the input: (len(units), N, 40, 40)."""
        net = self._net
        xd = _as_device_f32(x, net.device)
        outs = []
        for s in range(0, xd.shape[0], int(batch_size)):
            outs.append(net.attribute(xd[s:s + int(batch_size)], units, grad_x_input))
        return torch.cat(outs, dim=1).cpu().numpy()


__all__ = ["sample", "Encoder", "Decoder", "VAE", "Adam", "EarlyStopping", "History", "DeviceArray", "tf", "keras"]
