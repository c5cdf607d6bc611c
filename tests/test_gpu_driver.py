"""The caller side of the hot path with numbers checked against the oracle:
(1) one ``fit`` epoch's logged metrics = the oracle's unweighted running mean over the same permutation,
    partial last batch included, with the Philox epsilon restated on the host; validation included;
(2) the train.py-shaped driver (examples/train_driver.py, this package's own code over the compat module
    paths): grid x folds -> refit -> save_weights -> new model -> train_on_batch -> load_weights, then the
    reloaded model's outputs against the float64 oracle on the saved weights."""
import os
import sys

import numpy as np
import pytest

from oracle import vae_oracle as vo
from util import RTOL, rel_err

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# (switches, learning rate, tolerance): the alternate Keras readings through the driver.  Batch statistics make the
# loss chaotic in the step size (tests/test_gpu_net.py), hence the small rate and the wider bound there.
FIT_SEMS = [(None, 1e-3, 5e-5), ({"bce": "probs"}, 1e-3, 5e-5), ({"bn": "batch"}, 1e-5, 1e-3)]


@pytest.mark.parametrize("sem,lr,tol", FIT_SEMS, ids=["default", "bce=probs", "bn=batch"])
@pytest.mark.parametrize("variant", ["vae", "ae"])
def test_fit_epoch_metrics_equal_the_oracle_running_means(variant, sem, lr, tol):
    if variant == "vae":
        from vae_eeg_xai_amd.models import Encoder, Decoder, VAE as Model, Adam
    else:
        from vae_eeg_xai_amd.models_standard import (EncoderStandard as Encoder, DecoderStandard as Decoder,
                                                     Autoencoder as Model, Adam)
    N, NV, bs, L, seed, epochs = 20, 12, 8, 8, 3, 2        # batches of 8, 8, 4; validation 8, 4
    p = vo.init_params(L, variant, seed=4, jitter=0.1)
    x = vo.synth_topomaps(N + NV, 12)
    xt, xv = x[:N], x[N:]
    model = Model(Encoder(L), Decoder(), semantics=sem)
    model._net.set_weights_dict(p)
    model.compile(Adam(lr))
    hist = model.fit(xt, xt, bs, epochs, validation_data=(xv, xv), verbose=0, seed=seed)
    # validation (test_step) always normalises with the moving statistics, which bn="batch" keeps updating
    val_sem = dict(sem or {}, bn="inference")

    names = vo.trainable_names(L, variant)
    pp = {k: a.copy() for k, a in p.items()}
    m = {k: np.zeros_like(p[k]) for k in names}
    v = {k: np.zeros_like(p[k]) for k in names}
    rng = np.random.default_rng(seed)
    t = seen = 0
    keys = ("loss", "reconstruction_loss", "kl_loss") if variant == "vae" else ("loss",)
    for epoch in range(epochs):
        perm = rng.permutation(N)
        rows = []
        for s in range(0, N, bs):
            idx = perm[s:s + bs]
            eps = vo.philox_normal(42, seen, len(idx), L) if variant == "vae" else None
            seen += len(idx)
            t += 1
            rows.append(vo.train_step(pp, m, v, t, lr, xt[idx], eps, variant, sem))
        train_mean = np.mean(np.asarray(rows, dtype=np.float64), axis=0)         # unweighted over batches (Keras Mean)
        rows = []
        for s in range(0, NV, bs):
            xb = xv[s:s + bs]
            eps = vo.philox_normal(42, seen, len(xb), L) if variant == "vae" else None
            seen += len(xb)
            c = vo.forward(pp, xb, eps, variant, val_sem)
            rows.append((c["total"], c["rec_loss"], c["kl_loss"]))
        val_mean = np.mean(np.asarray(rows, dtype=np.float64), axis=0)
        scale = abs(train_mean[0])
        for i, k in enumerate(keys):
            # 5e-5 of the total loss: the device draws epsilon with fp32 log / sin / cos (2e-5 absolute of the float64
            # restatement, tests/test_gpu_net.py), and Adam at lr 1e-3 carries that into the later batches
            assert abs(hist.history[k][epoch] - train_mean[i]) < tol * scale, (epoch, k, hist.history[k][epoch], train_mean[i])
            assert abs(hist.history["val_" + k][epoch] - val_mean[i]) < tol * max(scale, abs(val_mean[0])), (epoch, k)
    assert model._net.samples_seen == seen


def test_train_py_shaped_driver(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import train_driver as td
    best, results, history, vae, vae2, x = td.main(str(tmp_path))
    assert len(results) == 2 and best in [r[0] for r in results]
    assert all(0.0 < s <= 1.0 for _, s in results)
    assert set(history.history) == {"loss", "reconstruction_loss", "kl_loss", "val_loss", "val_reconstruction_loss",
                                    "val_kl_loss"}
    assert len(history.history["loss"]) == 2
    # load_weights after the dummy train_on_batch (analysis.py:671-672) restores every array bit for bit
    for a, b in zip(vae.get_weights(), vae2.get_weights()):
        assert np.array_equal(a, b)
    # the reloaded model against the float64 oracle on the saved weights, same injected epsilon
    L, B = 8, 8
    w = {k: v.astype(np.float64) for k, v in vae2._net.get_weights_dict().items()}
    eps = np.random.default_rng(5).standard_normal((B, L)).astype(np.float32)
    c = vo.forward(w, x[:B], eps, "vae")
    assert rel_err(vae2.predict(x[:B], epsilon=eps), c["recon"]) < RTOL
    zm, zlv, z = vae2.encoder(x[:B], training=False, epsilon=eps)
    assert rel_err(zm, c["z_mean"]) < RTOL and rel_err(z, c["z"]) < RTOL
    vae2._net.metrics.zero_()
    logs = vae2.test_step((x[:B], x[:B]), epsilon=eps)
    assert abs(logs["loss"] - c["total"]) < RTOL * abs(c["total"])
    assert abs(logs["kl_loss"] - c["kl_loss"]) < RTOL * max(abs(c["kl_loss"]), 1.0)


@pytest.mark.parametrize("variant", ["vae", "ae"])
def test_fit_from_a_captured_graph_equals_the_eager_fit(variant):
    """VAE.fit replays full batches from a captured HIP graph (gather + train_fwd_bwd + Adam; the Philox sample
    counter is read from a device word at run time): history and final weights are bit-identical to the eager
    fit -- ragged last batch, validation between epochs and a change of the learning rate (new capture) included."""
    import torch
    if variant == "vae":
        from vae_eeg_xai_amd.models import Encoder, Decoder, VAE as Model, Adam
    else:
        from vae_eeg_xai_amd.models_standard import (EncoderStandard as Encoder, DecoderStandard as Decoder,
                                                     Autoencoder as Model, Adam)
    N, NV, bs, L = 150, 40, 32, 28            # batches 32, 32, 32, 32, 22
    p = vo.init_params(L, variant, seed=6, jitter=0.1)
    x = vo.synth_topomaps(N + NV, 21)
    out = []
    for graph in (False, True):
        model = Model(Encoder(L), Decoder())
        model._net.set_weights_dict(p)
        model.compile(Adam(1e-3))
        h1 = model.fit(x[:N], x[:N], bs, 3, validation_data=(x[N:], x[N:]), verbose=0, seed=5, graph=graph)
        model.compile(Adam(3e-4))
        h2 = model.fit(x[:N], x[:N], bs, 2, verbose=0, seed=6, graph=graph)
        torch.cuda.synchronize()
        assert ("_train_graphs" in model._net.__dict__) == graph
        out.append((h1.history, h2.history, model._net.params.clone(), model._net.adam_v.clone(), model._net.samples_seen))
    (a1, a2, pa, va, sa), (b1, b2, pb, vb, sb) = out
    assert a1 == b1 and a2 == b2 and sa == sb
    assert torch.equal(pa, pb) and torch.equal(va, vb)


def test_fit_on_the_split_fp32_engine_equals_the_fp32_mfma_fit():
    """VAE(..., compute="f32x3"): the same fit on the two fp32 engines (batch 128: the split kernel takes launches) --
    every logged metric within 1e-5 of the other engine's."""
    from vae_eeg_xai_amd.models import Encoder, Decoder, VAE, Adam
    x = vo.synth_topomaps(384 + 128, 21)
    hist = {}
    for compute in ("f32", "f32x3"):
        model = VAE(Encoder(28), Decoder(), compute=compute)
        assert model._net.compute == compute
        model.compile(Adam(1e-5))
        hist[compute] = model.fit(x[:384], x[:384], 128, 2, validation_data=(x[384:], x[384:]), verbose=0, seed=5).history
    for k, v in hist["f32"].items():
        for a, b in zip(v, hist["f32x3"][k]):
            assert abs(a - b) <= 1e-5 * max(abs(a), 1.0), (k, a, b)
