"""NumPy float64 restatement of the reference VAE / AE hot path (TEST INFRASTRUCTURE ONLY).

PARITY UNPINNED (see oracle/__init__.py): no TF/Keras here, no fixtures in the
reference.  Every function cites the reference line it follows; Keras-internal
semantics are marked [KERAS-SEM].

Formulation: explicit tap loops.  Conv2D is written in *gather* form, Conv2DTranspose
in *scatter* form; the backward pass is derived by hand (no autograd), so it is
independent of ``oracle/torch_ref.py`` which uses torch.nn.functional + autograd.
"""
from __future__ import annotations

import numpy as np

BN_EPS = 1e-3          # keras BatchNormalization default epsilon [KERAS-SEM]
BN_MOMENTUM = 0.99     # keras BatchNormalization default momentum [KERAS-SEM]
# keras.backend.epsilon(): clip and log offset of the probability-form BCE [KERAS-SEM].  Keras evaluates that form in
# float32, where the upper clip 1 - 1e-7 is not representable and rounds to 1 - 2**-23 = 1 - 1.19e-7: a saturated pixel
# costs -log(2.19e-7) = 15.33, not -log(2e-7) = 15.42, so the constants are restated as float32 holds them.
BCE_CLIP = float(np.float32(1e-7))
BCE_CLIP_HI = float(np.float32(1.0) - np.float32(1e-7))

# The two Keras semantics of the path that nothing in /root/reference can pin (SURVEY 7 "Hard parts", A.3, A.4), each
# with its alternative one switch away.  The DEFAULT is what the survey reads out of Keras 2.13 for models.py:
#   bn  = "inference": train_step calls self.encoder(data) / self.decoder(z) without ``training`` (models.py:152-153),
#         so BatchNormalization normalises with its moving statistics (frozen at 0 / 1) also inside fit();
#         "batch": the layer normalises with the statistics of the batch (biased variance), updates the moving
#         statistics with momentum 0.99, and gradients flow through the statistics (what training=True would do);
#   bce = "logits": keras.losses.binary_crossentropy finds the sigmoid's cached ``_keras_logits`` and evaluates
#         tf.nn.sigmoid_cross_entropy_with_logits; "probs": the fallback on the clipped probabilities,
#         -(t*log(p+1e-7) + (1-t)*log(1-p+1e-7)) with p = clip(sigmoid(a), 1e-7, 1-1e-7).
DEFAULT_SEM = {"bn": "inference", "bce": "logits"}
SEM_CHOICES = {"bn": ("inference", "batch"), "bce": ("logits", "probs", "probs64")}


def resolve_sem(sem=None):
    d = dict(DEFAULT_SEM)
    for k, v in (sem or {}).items():
        if k not in SEM_CHOICES or v not in SEM_CHOICES[k]:
            raise ValueError(f"unknown Keras-semantics switch {k}={v!r}; choices: {SEM_CHOICES}")
        d[k] = v
    return d
ADAM_B1, ADAM_B2, ADAM_EPS = 0.9, 0.999, 1e-7   # keras legacy Adam defaults [KERAS-SEM]

# ----------------------------------------------------------------------------
# Layer table (reference models.py:219-241 encoder, :263-301 decoder)
#   kind, name, Cin, Cout, k, stride, padding, has_bn
# ----------------------------------------------------------------------------
ENC_CONVS = [
    ("conv1", 1, 64, 3, 2, "same"),     # models.py:219-223
    ("conv2", 64, 128, 3, 2, "same"),   # models.py:225-229
    ("conv3", 128, 256, 3, 2, "same"),  # models.py:231-235
]
DEC_DECONVS = [
    ("deconv1", 256, 256, 3, 2, "same"),   # models.py:270-274
    ("deconv2", 256, 128, 3, 1, "same"),   # models.py:276-280
    ("deconv3", 128, 128, 3, 2, "valid"),  # models.py:282-286
    ("deconv4", 128, 64, 3, 1, "valid"),   # models.py:288-292
    ("deconv5", 64, 64, 3, 2, "valid"),    # models.py:294-298
    ("deconv6", 64, 1, 2, 1, "valid"),     # models.py:300-301 (sigmoid, no BN)
]


def _layer_order_spec(latent_dim: int, variant: str = "vae"):
    """(name, shape, trainable) in layer-creation order, trainable before non-trainable.  init_params
    draws its random numbers in THIS order (the committed golden fixtures depend on it)."""
    L = latent_dim
    tr, ntr = [], []

    def bn(prefix, c):
        tr.append((prefix + ".gamma", (c,)))
        tr.append((prefix + ".beta", (c,)))
        ntr.append((prefix + ".mean", (c,)))
        ntr.append((prefix + ".var", (c,)))

    for name, cin, cout, k, s, pad in ENC_CONVS:
        tr.append((f"enc.{name}.kernel", (k, k, cin, cout)))
        tr.append((f"enc.{name}.bias", (cout,)))
        bn(f"enc.{name}", cout)
    if variant == "vae":
        tr.append(("enc.dense.kernel", (6400, 100)))     # models.py:238
        tr.append(("enc.dense.bias", (100,)))
        tr.append(("enc.z_mean.kernel", (100, L)))        # models.py:240
        tr.append(("enc.z_mean.bias", (L,)))
        tr.append(("enc.z_log_var.kernel", (100, L)))     # models.py:241
        tr.append(("enc.z_log_var.bias", (L,)))
    else:
        tr.append(("enc.dense.kernel", (6400, L)))        # models_standard.py:108
        tr.append(("enc.dense.bias", (L,)))
    tr.append(("dec.dense.kernel", (L, 4096)))            # models.py:264
    tr.append(("dec.dense.bias", (4096,)))
    bn("dec.dense", 4096)                                 # models.py:265
    for name, cin, cout, k, s, pad in DEC_DECONVS:
        tr.append((f"dec.{name}.kernel", (k, k, cout, cin)))   # Conv2DTranspose kernel (kh,kw,Cout,Cin)
        tr.append((f"dec.{name}.bias", (cout,)))
        if name != "deconv6":
            bn(f"dec.{name}", cout)
    return [(n, s, True) for n, s in tr] + [(n, s, False) for n, s in ntr]


def param_spec(latent_dim: int, variant: str = "vae"):
    """Ordered (name, shape, trainable) list in Keras ``Model.get_weights()`` order (train.py:287,
    analysis.py:127) -- SURVEY A.6 [KERAS-SEM], PARITY UNPINNED.  ``Model.weights`` concatenates
    ``layer.variables`` of the tracked sub-layers (encoder, decoder, then the three metric trackers,
    models.py:125-133), and a ``Layer``'s variables are its trainable weights in creation order
    followed by its non-trainable ones: encoder trainable, encoder BN moving statistics, decoder
    trainable, decoder BN moving statistics.  (The trackers' total/count scalars are not listed.)"""
    spec = _layer_order_spec(latent_dim, variant)
    out = []
    for part in ("enc.", "dec."):
        out += [t for t in spec if t[0].startswith(part) and t[2]]
        out += [t for t in spec if t[0].startswith(part) and not t[2]]
    return out


def init_params(latent_dim: int, variant: str = "vae", seed: int = 42, jitter: float = 0.0):
    """Random-init weights of the reference architecture.

    he_uniform for conv kernels: limit sqrt(6/fan_in); Keras computes fan_in for a 4-D
    kernel as prod(shape[:-2]) * shape[-2], i.e. kh*kw*Cin for Conv2D and kh*kw*Cout for
    Conv2DTranspose [KERAS-SEM].  Dense: glorot_uniform.  Biases 0, BN gamma=1 beta=0
    mean=0 var=1.  ``jitter`` > 0 perturbs biases / BN tensors so that parity tests do not
    pass trivially on zeros / ones.
    """
    rng = np.random.default_rng(seed)
    p = {}
    for name, shape, trainable in _layer_order_spec(latent_dim, variant):
        leaf = name.rsplit(".", 1)[1]
        if leaf == "kernel":
            if len(shape) == 4:
                fan_in = shape[0] * shape[1] * shape[2]
                lim = np.sqrt(6.0 / fan_in)
            else:
                lim = np.sqrt(6.0 / (shape[0] + shape[1]))
            v = rng.uniform(-lim, lim, size=shape)
        elif leaf in ("bias", "beta", "mean"):
            v = np.zeros(shape)
            if jitter:
                v = v + jitter * rng.standard_normal(shape)
        else:  # gamma, var
            v = np.ones(shape)
            if jitter:
                v = v * np.exp(jitter * rng.standard_normal(shape))
        p[name] = np.asarray(v, dtype=np.float64)
    return p


def synth_topomaps(n: int, seed: int = 1234) -> np.ndarray:
    """Synthetic (n,40,40,1) float32 maps in [0,1] imitating the reference's input
    (indian_functions.py:112-124 cubic field inside the hull, 0 outside; train.py:95-100
    expand + global min-max).  SURVEY 8d generator."""
    from scipy.ndimage import zoom

    rng = np.random.default_rng(seed)
    coarse = rng.standard_normal((n, 8, 8))
    fine = zoom(coarse, (1, 5, 5), order=3)
    yy, xx = np.mgrid[0:40, 0:40]
    disc = ((yy - 19.5) ** 2 + (xx - 19.5) ** 2) <= 19.5 ** 2
    fine = fine * disc[None]
    x = np.expand_dims(fine, -1).astype("float32")           # train.py:95-96
    x = (x - x.min()) / (x.max() - x.min())                    # train.py:99-100
    return x.astype(np.float32)


# ----------------------------------------------------------------------------
# primitive ops (forward + hand-derived backward)
# ----------------------------------------------------------------------------
def same_pad(n_in, k, s):
    """TF 'SAME' padding arithmetic [KERAS-SEM] -> (out, pad_before, pad_after)."""
    out = -(-n_in // s)
    tot = max((out - 1) * s + k - n_in, 0)
    return out, tot // 2, tot - tot // 2


def conv2d_fwd(x, W, b, stride, padding):
    """Conv2D NHWC, kernel (kh,kw,Cin,Cout), cross-correlation (models.py:220-221). Gather form."""
    n, H, Wd, cin = x.shape
    kh, kw, _, cout = W.shape
    if padding == "same":
        Ho, pt, pb_ = same_pad(H, kh, stride)
        Wo, pl, pr = same_pad(Wd, kw, stride)
    else:
        Ho, pt, pb_ = (H - kh) // stride + 1, 0, 0
        Wo, pl, pr = (Wd - kw) // stride + 1, 0, 0
    xp = np.pad(x, ((0, 0), (pt, pb_), (pl, pr), (0, 0)))
    y = np.zeros((n, Ho, Wo, cout))
    for ky in range(kh):
        for kx in range(kw):
            patch = xp[:, ky:ky + stride * Ho:stride, kx:kx + stride * Wo:stride, :]
            y += patch @ W[ky, kx]
    return y + b


def conv2d_bwd(x, W, dy, stride, padding):
    """Returns dx, dW, db for conv2d_fwd."""
    n, H, Wd, cin = x.shape
    kh, kw, _, cout = W.shape
    _, Ho, Wo, _ = dy.shape
    if padding == "same":
        _, pt, pb_ = same_pad(H, kh, stride)
        _, pl, pr = same_pad(Wd, kw, stride)
    else:
        pt = pb_ = pl = pr = 0
    xp = np.pad(x, ((0, 0), (pt, pb_), (pl, pr), (0, 0)))
    dxp = np.zeros_like(xp)
    dW = np.zeros_like(W)
    for ky in range(kh):
        for kx in range(kw):
            sl = (slice(None), slice(ky, ky + stride * Ho, stride), slice(kx, kx + stride * Wo, stride), slice(None))
            dW[ky, kx] = np.einsum("nhwi,nhwo->io", xp[sl], dy)
            dxp[sl] += dy @ W[ky, kx].T
    dx = dxp[:, pt:pt + H, pl:pl + Wd, :]
    return dx, dW, dy.sum(axis=(0, 1, 2))


def deconv_geometry(n_in, k, s, padding):
    """Conv2DTranspose output length and crop-at-start [KERAS-SEM]."""
    full = (n_in - 1) * s + k
    if padding == "valid":
        out = n_in * s + max(k - s, 0)
        return out, 0, full
    out = n_in * s
    _, pbefore, _ = same_pad(out, k, s)
    return out, pbefore, full


def deconv2d_fwd(x, W, b, stride, padding):
    """Conv2DTranspose NHWC, kernel (kh,kw,Cout,Cin) (models.py:271-301). Scatter form:
    y[n, s*iy+ky-pb, s*ix+kx-pb, co] += x[n,iy,ix,ci] * W[ky,kx,co,ci]."""
    n, H, Wd, cin = x.shape
    kh, kw, cout, _ = W.shape
    Ho, pt, fullh = deconv_geometry(H, kh, stride, padding)
    Wo, pl, fullw = deconv_geometry(Wd, kw, stride, padding)
    full = np.zeros((n, max(fullh, pt + Ho), max(fullw, pl + Wo), cout))
    for ky in range(kh):
        for kx in range(kw):
            full[:, ky:ky + stride * H:stride, kx:kx + stride * Wd:stride, :] += x @ W[ky, kx].T
    return full[:, pt:pt + Ho, pl:pl + Wo, :] + b


def deconv2d_bwd(x, W, dy, stride, padding):
    n, H, Wd, cin = x.shape
    kh, kw, cout, _ = W.shape
    Ho, pt, fullh = deconv_geometry(H, kh, stride, padding)
    Wo, pl, fullw = deconv_geometry(Wd, kw, stride, padding)
    dfull = np.zeros((n, max(fullh, pt + Ho), max(fullw, pl + Wo), cout))
    dfull[:, pt:pt + Ho, pl:pl + Wo, :] = dy
    dx = np.zeros_like(x)
    dW = np.zeros_like(W)
    for ky in range(kh):
        for kx in range(kw):
            g = dfull[:, ky:ky + stride * H:stride, kx:kx + stride * Wd:stride, :]   # (n,H,W,cout)
            dx += g @ W[ky, kx]
            dW[ky, kx] = np.einsum("nhwo,nhwi->oi", g, x)
    return dx, dW, dy.sum(axis=(0, 1, 2))


def bn_fwd(a, gamma, beta, mean, var):
    """BatchNormalization in inference mode (SURVEY A.3 [KERAS-SEM]): the reference never
    forwards ``training`` (models.py:152-153,244-253,303-312)."""
    rstd = 1.0 / np.sqrt(var + BN_EPS)
    return (a - mean) * rstd * gamma + beta


def bn_bwd(a, gamma, mean, var, dy):
    rstd = 1.0 / np.sqrt(var + BN_EPS)
    axes = tuple(range(dy.ndim - 1))
    dgamma = (dy * (a - mean) * rstd).sum(axis=axes)
    dbeta = dy.sum(axis=axes)
    return dy * gamma * rstd, dgamma, dbeta


def bn_fwd_batch(a, gamma, beta):
    """BatchNormalization with training=True [KERAS-SEM]: batch mean and BIASED variance over every axis but the
    last.  Returns (y, mean, var)."""
    axes = tuple(range(a.ndim - 1))
    mu = a.mean(axis=axes)
    var = a.var(axis=axes)
    rstd = 1.0 / np.sqrt(var + BN_EPS)
    return (a - mu) * rstd * gamma + beta, mu, var


def bn_bwd_batch(a, gamma, mu, var, dy):
    """Backward of bn_fwd_batch THROUGH the statistics: da = rstd * (g - mean(g) - xhat * mean(g * xhat)),
    g = dy * gamma."""
    axes = tuple(range(dy.ndim - 1))
    rstd = 1.0 / np.sqrt(var + BN_EPS)
    xhat = (a - mu) * rstd
    dgamma = (dy * xhat).sum(axis=axes)
    dbeta = dy.sum(axis=axes)
    g = dy * gamma
    da = rstd * (g - g.mean(axis=axes) - xhat * (g * xhat).mean(axis=axes))
    return da, dgamma, dbeta


def bn_moving_update(moving_mean, moving_var, mu, var, n_reduced, four_d):
    """Moving statistics after one training=True call [KERAS-SEM]: moving <- moving * 0.99 + batch * 0.01; the 4-D
    (fused) layers feed the UNBIASED batch variance (n / (n - 1)), the 2-D BatchNormalization after dec.dense the
    biased one."""
    v = var * (n_reduced / max(n_reduced - 1, 1)) if four_d else var
    return (moving_mean * BN_MOMENTUM + mu * (1 - BN_MOMENTUM), moving_var * BN_MOMENTUM + v * (1 - BN_MOMENTUM))


def _bn_apply(c, p, prefix, a, sem):
    """The BatchNormalization after layer ``prefix`` in the selected mode; caches what the backward needs and, in
    batch mode, the moving statistics the step leaves behind (c["bn_state"])."""
    if sem["bn"] == "inference":
        return bn_fwd(a, p[prefix + ".gamma"], p[prefix + ".beta"], p[prefix + ".mean"], p[prefix + ".var"])
    y, mu, var = bn_fwd_batch(a, p[prefix + ".gamma"], p[prefix + ".beta"])
    c[prefix + ".bn_mu"], c[prefix + ".bn_var"] = mu, var
    n_red = a.size // a.shape[-1]
    mm, mv = bn_moving_update(p[prefix + ".mean"], p[prefix + ".var"], mu, var, n_red, a.ndim == 4)
    c.setdefault("bn_state", {})[prefix + ".mean"] = mm
    c["bn_state"][prefix + ".var"] = mv
    return y


def _bn_back(c, p, prefix, a, dy, sem):
    if sem["bn"] == "inference":
        return bn_bwd(a, p[prefix + ".gamma"], p[prefix + ".mean"], p[prefix + ".var"], dy)
    return bn_bwd_batch(a, p[prefix + ".gamma"], c[prefix + ".bn_mu"], c[prefix + ".bn_var"], dy)


def _sigmoid_f32(a):
    a = np.asarray(a, dtype=np.float32)
    e = np.exp(-np.abs(a))
    one = np.float32(1.0)
    return np.where(a >= 0, one / (one + e), e / (one + e)).astype(np.float32)


def bce_probs(a, t, f32=True):
    """keras.backend.binary_crossentropy on probabilities [KERAS-SEM] (the fallback form).

    Keras evaluates this form in float32 and the form is NOT continuous there: sigmoid saturates to exactly 1 at
    a > 16.6, 1 - p is a multiple of 2**-24, and the clip cuts at 1 - 2**-23.  A float64 evaluation is a different
    function of the logits wherever |a| > ~12 (the gradient jumps from 0.54 to 0 across the clip edge), so the
    restatement follows the float32 arithmetic (``f32=True``, what sem "probs" uses); ``f32=False`` ("probs64") is the
    same formula in real arithmetic, for the finite-difference / autograd checks of the formula itself."""
    if not f32:
        pr = np.clip(sigmoid(a), BCE_CLIP, BCE_CLIP_HI)
        return -(t * np.log(pr + BCE_CLIP) + (1.0 - t) * np.log(1.0 - pr + BCE_CLIP))
    ce, hi, one = np.float32(BCE_CLIP), np.float32(BCE_CLIP_HI), np.float32(1.0)
    t = np.asarray(t, dtype=np.float32)
    pr = np.clip(_sigmoid_f32(a), ce, hi)
    return (-(t * np.log(pr + ce) + (one - t) * np.log(one - pr + ce))).astype(np.float64)


def bce_probs_dlogit(a, t, f32=True):
    """d bce_probs / d a: zero where the clip is active (tf.clip_by_value passes no gradient outside its range)."""
    if not f32:
        sg = sigmoid(a)
        inside = (sg >= BCE_CLIP) & (sg <= BCE_CLIP_HI)
        pr = np.clip(sg, BCE_CLIP, BCE_CLIP_HI)
        dpr = -t / (pr + BCE_CLIP) + (1.0 - t) / (1.0 - pr + BCE_CLIP)
        return dpr * sg * (1.0 - sg) * inside
    ce, hi, one = np.float32(BCE_CLIP), np.float32(BCE_CLIP_HI), np.float32(1.0)
    t = np.asarray(t, dtype=np.float32)
    sg = _sigmoid_f32(a)
    inside = (sg >= ce) & (sg <= hi)
    pr = np.clip(sg, ce, hi)
    dpr = (one - t) / (one - pr + ce) - t / (pr + ce)
    return (dpr * sg * (one - sg) * inside).astype(np.float64)


# |logit| beyond which float32 rounding of 1 - sigmoid decides the value of the probability form (1 - p < 6e-6):
# parity tests take the device's own d loss / d logit there (``backward(..., masks={"dlogits": ...})``), as they take
# its ReLU decisions at the kinks.
BCE_PROBS_BAND = 12.0


def bce_logits(a, t):
    """tf.nn.sigmoid_cross_entropy_with_logits, reached via the _keras_logits shortcut
    of keras.losses.binary_crossentropy (models.py:156-160) [KERAS-SEM]."""
    return np.maximum(a, 0) - a * t + np.log1p(np.exp(-np.abs(a)))


def sigmoid(a):
    return 1.0 / (1.0 + np.exp(-a))


# ----------------------------------------------------------------------------
# model forward / backward
# ----------------------------------------------------------------------------
def encoder_fwd(p, x, variant="vae", sem=None):
    """Encoder.call (models.py:244-253) / EncoderStandard.call (models_standard.py:110-116).
    Layer order Conv(+bias) -> ReLU -> BN (models.py:219-223)."""
    sem = resolve_sem(sem)
    c = {"x": x}
    h = x
    for name, cin, cout, k, s, pad in ENC_CONVS:
        pre = conv2d_fwd(h, p[f"enc.{name}.kernel"], p[f"enc.{name}.bias"], s, pad)
        a = np.maximum(pre, 0)
        h = _bn_apply(c, p, f"enc.{name}", a, sem)
        c[f"enc.{name}.pre"] = pre
        c[f"enc.{name}.a"] = a
        c[f"enc.{name}.y"] = h
    flat = h.reshape(h.shape[0], -1)                              # Flatten (h,w,c) models.py:237
    c["enc.flat"] = flat
    dpre_ = flat @ p["enc.dense.kernel"] + p["enc.dense.bias"]
    d = np.maximum(dpre_, 0)   # Dense relu
    c["enc.dense.pre"] = dpre_
    c["enc.dense.a"] = d
    if variant == "vae":
        c["z_mean"] = d @ p["enc.z_mean.kernel"] + p["enc.z_mean.bias"]
        c["z_log_var"] = d @ p["enc.z_log_var.kernel"] + p["enc.z_log_var.bias"]
    return c


def sample(z_mean, z_log_var, eps):
    """sample() (models.py:10-22) with an injected epsilon."""
    return z_mean + np.exp(0.5 * z_log_var) * eps


def decoder_fwd(p, z, c=None, sem=None):
    """Decoder.call (models.py:303-312): Dense relu -> BN -> reshape (4,4,256) -> 5x[deconv relu BN] -> deconv6."""
    sem = resolve_sem(sem)
    c = {} if c is None else c
    c["z"] = z
    pre = z @ p["dec.dense.kernel"] + p["dec.dense.bias"]
    a = np.maximum(pre, 0)
    c["dec.dense.pre"] = pre
    c["dec.dense.a"] = a
    h = _bn_apply(c, p, "dec.dense", a, sem)
    h = h.reshape(-1, 4, 4, 256)
    for name, cin, cout, k, s, pad in DEC_DECONVS:
        c[f"dec.{name}.in"] = h
        pre = deconv2d_fwd(h, p[f"dec.{name}.kernel"], p[f"dec.{name}.bias"], s, pad)
        if name == "deconv6":
            c["logits"] = pre
            c["recon"] = sigmoid(pre)
            break
        a = np.maximum(pre, 0)
        c[f"dec.{name}.pre"] = pre
        c[f"dec.{name}.a"] = a
        h = _bn_apply(c, p, f"dec.{name}", a, sem)
    return c


def forward(p, x, eps=None, variant="vae", sem=None):
    """VAE.train_step forward half (models.py:152-163) / Autoencoder (models_standard.py:36-44).
    Returns cache with all intermediates plus rec_loss, kl_loss, total."""
    sem = resolve_sem(sem)
    x = np.asarray(x, dtype=np.float64)
    c = encoder_fwd(p, x, variant, sem)
    c["sem"] = sem
    if variant == "vae":
        c["eps"] = np.asarray(eps, dtype=np.float64)
        z = sample(c["z_mean"], c["z_log_var"], c["eps"])
    else:
        z = c["enc.dense.a"]
    decoder_fwd(p, z, c, sem)
    B = x.shape[0]
    if sem["bce"] == "logits":
        bce = bce_logits(c["logits"], x)
    else:
        bce = bce_probs(c["logits"], x, f32=(sem["bce"] == "probs"))
    bce = bce.mean(axis=-1)                                       # mean over last axis (size 1)
    c["rec_per_sample"] = bce.sum(axis=(1, 2))
    c["rec_loss"] = c["rec_per_sample"].mean()
    if variant == "vae":
        mu, lv = c["z_mean"], c["z_log_var"]
        c["kl_per_sample"] = (-0.5 * (1 + lv - mu ** 2 - np.exp(lv))).sum(axis=1)   # models.py:161-162
        c["kl_loss"] = c["kl_per_sample"].mean()
    else:
        c["kl_loss"] = 0.0
    c["total"] = c["rec_loss"] + c["kl_loss"]
    return c


def backward(p, c, variant="vae", masks=None):
    """Gradients of ``total`` wrt every trainable tensor (tape.gradient, models.py:166).

    ``masks`` (optional) maps "<layer>" (e.g. "dec.deconv3", "enc.dense") to a boolean ReLU
    mask that replaces ``a > 0``: a pre-activation within fp32 rounding of 0 has an ambiguous
    sub-gradient, so parity tests evaluate the oracle at the device's ReLU decisions after
    checking that every disagreement sits at such a kink.  ``masks["dlogits"]`` (optional) replaces d total /
    d logits the same way, for the saturation band of the float32 probability form (``BCE_PROBS_BAND``)."""
    g = {}
    masks = masks or {}
    sem = c.get("sem", resolve_sem(None))       # the switches the forward cache was made with

    def on(layer, a):
        return masks[layer] if layer in masks else (a > 0)
    x = c["x"]
    B = x.shape[0]
    if sem["bce"] == "logits":
        dh = (sigmoid(c["logits"]) - x) / B                       # d rec / d logits
    else:
        dh = bce_probs_dlogit(c["logits"], x, f32=(sem["bce"] == "probs")) / B
    if "dlogits" in masks:                                        # the caller's d loss / d logits (BCE_PROBS_BAND)
        dh = np.asarray(masks["dlogits"], dtype=np.float64).reshape(dh.shape)
    g["dlogits"] = dh
    # decoder backwards
    for name, cin, cout, k, s, pad in reversed(DEC_DECONVS):
        if name != "deconv6":
            a = c[f"dec.{name}.a"]
            da, g[f"dec.{name}.gamma"], g[f"dec.{name}.beta"] = _bn_back(c, p, f"dec.{name}", a, dh, sem)
            dh = da * on(f"dec.{name}", a)
        g[f"dpre.dec.{name}"] = dh
        dh, g[f"dec.{name}.kernel"], g[f"dec.{name}.bias"] = deconv2d_bwd(
            c[f"dec.{name}.in"], p[f"dec.{name}.kernel"], dh, s, pad)
    dh = dh.reshape(B, 4096)
    a = c["dec.dense.a"]
    da, g["dec.dense.gamma"], g["dec.dense.beta"] = _bn_back(c, p, "dec.dense", a, dh, sem)
    dpre = da * on("dec.dense", a)
    g["dpre.dec.dense"] = dpre
    g["dec.dense.kernel"] = c["z"].T @ dpre
    g["dec.dense.bias"] = dpre.sum(0)
    dz = dpre @ p["dec.dense.kernel"].T
    g["dz"] = dz
    d = c["enc.dense.a"]
    if variant == "vae":
        mu, lv, eps = c["z_mean"], c["z_log_var"], c["eps"]
        dmu = dz + mu / B
        dlv = dz * 0.5 * np.exp(0.5 * lv) * eps + 0.5 * (np.exp(lv) - 1) / B
        g["dz_mean"], g["dz_log_var"] = dmu, dlv
        g["enc.z_mean.kernel"] = d.T @ dmu
        g["enc.z_mean.bias"] = dmu.sum(0)
        g["enc.z_log_var.kernel"] = d.T @ dlv
        g["enc.z_log_var.bias"] = dlv.sum(0)
        dd = dmu @ p["enc.z_mean.kernel"].T + dlv @ p["enc.z_log_var.kernel"].T
    else:
        dd = dz
    dpre = dd * on("enc.dense", d)
    g["dpre.enc.dense"] = dpre
    g["enc.dense.kernel"] = c["enc.flat"].T @ dpre
    g["enc.dense.bias"] = dpre.sum(0)
    dh = (dpre @ p["enc.dense.kernel"].T).reshape(B, 5, 5, 256)
    for name, cin, cout, k, s, pad in reversed(ENC_CONVS):
        a = c[f"enc.{name}.a"]
        da, g[f"enc.{name}.gamma"], g[f"enc.{name}.beta"] = _bn_back(c, p, f"enc.{name}", a, dh, sem)
        dpre = da * on(f"enc.{name}", a)
        g[f"dpre.enc.{name}"] = dpre
        xin = c["x"] if name == "conv1" else c[f"enc.conv{int(name[-1]) - 1}.y"]
        dh, g[f"enc.{name}.kernel"], g[f"enc.{name}.bias"] = conv2d_bwd(
            xin, p[f"enc.{name}.kernel"], dpre, s, pad)
    g["dx"] = dh
    return g


def attribution(p, x, units, variant="vae", grad_x_input=False, masks=None):
    """Saliency d z_mean[:, j] / d x for each j in ``units`` (north_star item 3; no reference
    call site -- SURVEY 3.4).  For the AE variant the target is the bottleneck unit.
    ``masks``: optional ReLU decisions, see ``backward``.  Returns (len(units), B, 40, 40, 1)."""
    x = np.asarray(x, dtype=np.float64)
    masks = masks or {}

    def on(layer, a):
        return masks[layer] if layer in masks else (a > 0)

    c = encoder_fwd(p, x, variant)
    B = x.shape[0]
    out = []
    d = c["enc.dense.a"]
    for j in units:
        if variant == "vae":
            dd = np.tile(p["enc.z_mean.kernel"][:, j][None, :], (B, 1))
            dpre = dd * on("enc.dense", d)
        else:
            dpre = np.zeros_like(d)
            dpre[:, j] = on("enc.dense", d)[:, j]
        dh = (dpre @ p["enc.dense.kernel"].T).reshape(B, 5, 5, 256)
        for name, cin, cout, k, s, pad in reversed(ENC_CONVS):
            a = c[f"enc.{name}.a"]
            da, _, _ = bn_bwd(a, p[f"enc.{name}.gamma"], p[f"enc.{name}.mean"], p[f"enc.{name}.var"], dh)
            dpre = da * on(f"enc.{name}", a)
            xin = c["x"] if name == "conv1" else c[f"enc.conv{int(name[-1]) - 1}.y"]
            dh, _, _ = conv2d_bwd(xin, p[f"enc.{name}.kernel"], dpre, s, pad)
        out.append(dh * x if grad_x_input else dh)
    return np.stack(out)


def adam_step(params, grads, m, v, t, lr):
    """keras.optimizers.legacy.Adam (train.py:9,137) [KERAS-SEM]: epsilon is
    NOT divided by sqrt(1-beta2^t).  ``t`` is the 1-based step.  In-place on dicts."""
    alpha = lr * np.sqrt(1 - ADAM_B2 ** t) / (1 - ADAM_B1 ** t)
    for k in grads:
        if k not in params or k not in m:
            continue
        m[k] = ADAM_B1 * m[k] + (1 - ADAM_B1) * grads[k]
        v[k] = ADAM_B2 * v[k] + (1 - ADAM_B2) * grads[k] ** 2
        params[k] = params[k] - alpha * m[k] / (np.sqrt(v[k]) + ADAM_EPS)


def trainable_names(latent_dim, variant="vae"):
    return [n for n, s, t in param_spec(latent_dim, variant) if t]


def train_step(p, m, v, t, lr, x, eps=None, variant="vae", sem=None):
    """One VAE.train_step (models.py:148-180): returns (total, rec, kl); updates p, m, v in place (and, with
    sem bn="batch", the BatchNormalization moving statistics)."""
    c = forward(p, x, eps, variant, sem)
    g = backward(p, c, variant)
    names = trainable_names(p["dec.dense.kernel"].shape[0], variant)
    adam_step(p, {k: g[k] for k in names}, m, v, t, lr)
    for k, val in c.get("bn_state", {}).items():
        p[k] = val
    return c["total"], c["rec_loss"], c["kl_loss"]


# ----------------------------------------------------------------------------
# Philox-4x32-10 + Box-Muller (the device epsilon generator restated; north_star item 2)
# ----------------------------------------------------------------------------
_PH_M0, _PH_M1 = 0xD2511F53, 0xCD9E8D57
_PH_W0, _PH_W1 = 0x9E3779B9, 0xBB67AE85


def philox4x32(counter, key):
    """counter: (n,4) uint32, key: (2,) uint32 -> (n,4) uint32 (Salmon et al. 2011, 10 rounds)."""
    c = np.array(counter, dtype=np.uint64)
    k0, k1 = int(key[0]), int(key[1])
    for _ in range(10):
        p0 = _PH_M0 * c[:, 0]
        p1 = _PH_M1 * c[:, 2]
        hi0, lo0 = p0 >> 32, p0 & 0xFFFFFFFF
        hi1, lo1 = p1 >> 32, p1 & 0xFFFFFFFF
        c = np.stack([hi1 ^ c[:, 1] ^ k0, lo1, hi0 ^ c[:, 3] ^ k1, lo0], axis=1) & 0xFFFFFFFF
        k0 = (k0 + _PH_W0) & 0xFFFFFFFF
        k1 = (k1 + _PH_W1) & 0xFFFFFFFF
    return c.astype(np.uint32)


def philox_normal(seed: int, first_sample: int, n_samples: int, L: int):
    """Standard normals laid out (n_samples, L).  Element index e = (first_sample+b)*L + l
    uses Philox counter (e>>2 low, e>>34 high, 0, 0), key (seed lo, seed hi); lanes (0,1) and
    (2,3) of the 4x32 output feed two Box-Muller pairs; element e takes output e & 3.
    Independent of world size / batch split."""
    e = (np.arange(n_samples * L, dtype=np.uint64) + np.uint64(first_sample) * np.uint64(L))
    q = e >> np.uint64(2)
    ctr = np.stack([q & np.uint64(0xFFFFFFFF), q >> np.uint64(32),
                    np.zeros_like(q), np.zeros_like(q)], axis=1)
    r = philox4x32(ctr, (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)).astype(np.float64)
    u = (r + 0.5) * (1.0 / 4294967296.0)          # (0,1)
    rad0 = np.sqrt(-2.0 * np.log(u[:, 0]))
    rad1 = np.sqrt(-2.0 * np.log(u[:, 2]))
    th0 = 2.0 * np.pi * u[:, 1]
    th1 = 2.0 * np.pi * u[:, 3]
    n4 = np.stack([rad0 * np.cos(th0), rad0 * np.sin(th0), rad1 * np.cos(th1), rad1 * np.sin(th1)], axis=1)
    sel = (e & np.uint64(3)).astype(np.int64)
    out = n4[np.arange(len(e)), sel]
    return out.reshape(n_samples, L)
