"""GPU tests of the reference-shaped Python API (models.py / models_standard.py counterparts):
same call patterns as train.py / analysis.py use."""
import os

import numpy as np
import pytest

from oracle import vae_oracle as vo
from util import rel_err

pytestmark = pytest.mark.gpu


def test_vae_api_like_train_py(tmp_path):
    from vae_eeg_xai_amd.models import Encoder, Decoder, VAE, Adam, EarlyStopping, sample
    x = vo.synth_topomaps(200, 5)
    encoder = Encoder(28)
    decoder = Decoder()
    vae = VAE(encoder, decoder, 3, 1e-3, 64, 2)
    vae.compile(Adam(1e-3))
    es = EarlyStopping("val_loss", patience=2, verbose=0)
    hist = vae.fit(x[:160], x[:160], 64, 3, validation_data=(x[160:], x[160:]), callbacks=[es], verbose=0)
    assert set(hist.history) == {"loss", "reconstruction_loss", "kl_loss", "val_loss", "val_reconstruction_loss", "val_kl_loss"}
    assert len(hist.history["loss"]) == 3
    assert hist.history["loss"][-1] < hist.history["loss"][0]
    assert abs(hist.history["loss"][0] - hist.history["reconstruction_loss"][0] - hist.history["kl_loss"][0]) < 1e-2
    pred = vae.predict(x[160:])
    assert pred.shape == (40, 40, 40, 1) and pred.dtype == np.float32 and 0 <= pred.min() and pred.max() <= 1
    # encoder / decoder direct calls as analysis.py does (analysis.py:36,49,215,559,580)
    z_mean, z_log_var, z = vae.encoder(x[:10], training=False)
    assert z_mean.shape == (10, 28) and np.quantile(z_mean, 0.25, axis=0).shape == (28,)
    zc = np.copy(z)
    zc[:, 3] = np.median(zc[:, 3])
    out = vae.decoder(zc, training=False)
    assert out.shape == (10, 40, 40, 1) and out[0].shape == (40, 40, 1)
    one = vae.decoder(np.expand_dims(zc[0], 0))
    assert rel_err(one[0], out[0]) < 1e-6
    # injected epsilon reproduces sample()
    eps = np.random.default_rng(0).standard_normal((10, 28)).astype(np.float32)
    zm, zlv, z2 = vae.encoder(x[:10], epsilon=eps)
    want = np.asarray(zm) + np.exp(0.5 * np.asarray(zlv)) * eps
    assert rel_err(z2, want) < 1e-6
    assert rel_err(sample(zm, zlv, epsilon=eps), want) < 1e-6
    # weights: keras order, save / load round trip
    w = vae.get_weights()
    assert len(w) == 62 and w[0].shape == (3, 3, 1, 64)
    path = os.path.join(tmp_path, "vae_s10")
    vae.save_weights(path, save_format="tf")
    enc2, dec2 = Encoder(28), Decoder()
    vae2 = VAE(enc2, dec2)
    vae2.compile(Adam(1e-5))
    vae2.train_on_batch(x[:1], x[:1])              # analysis.py:671
    vae2.load_weights(path)
    for a, b in zip(vae2.get_weights(), w):
        assert np.array_equal(a, b)
    r1 = vae.predict(x[:8], epsilon=eps[:8])
    r2 = vae2.predict(x[:8], epsilon=eps[:8])
    assert np.array_equal(r1, r2)
    assert vae.train_step((x[:32], x[:32]))["loss"] > 0
    assert vae.test_step((x[:32], x[:32]))["kl_loss"] >= 0


def test_autoencoder_api_like_train_standard_py():
    from vae_eeg_xai_amd.models_standard import Autoencoder, EncoderStandard, DecoderStandard, Adam
    x = vo.synth_topomaps(96, 6)
    ae = Autoencoder(EncoderStandard(28), DecoderStandard(), 2, 1e-3, 32, 30)
    ae.compile(Adam(1e-3))
    hist = ae.fit(x, x, 32, 2, validation_data=(x[:32], x[:32]), verbose=0)
    assert set(hist.history) == {"loss", "val_loss"}
    assert hist.history["loss"][1] < hist.history["loss"][0]
    b = ae.encoder(x[:5])
    assert b.shape == (5, 28) and np.asarray(b).min() >= 0
    assert ae.decoder(b).shape == (5, 40, 40, 1)
    assert ae.predict(x[:5]).shape == (5, 40, 40, 1)
    assert len(ae.metrics) == 1


def test_early_stopping_stops():
    from vae_eeg_xai_amd.models import Encoder, Decoder, VAE, Adam, EarlyStopping
    x = vo.synth_topomaps(64, 9)
    vae = VAE(Encoder(8), Decoder())
    vae.compile(Adam(0.0))                         # lr 0: val_loss never improves after epoch 1
    es = EarlyStopping("val_loss", patience=2)
    eps_free = vae.fit(x, x, 32, 50, validation_data=(x, x), callbacks=[es], verbose=0)
    assert len(eps_free.history["loss"]) < 50


def test_attribute_api():
    from vae_eeg_xai_amd.models import Encoder, Decoder, VAE
    x = vo.synth_topomaps(10, 4)
    vae = VAE(Encoder(28), Decoder())
    maps = vae.attribute(x, [1, 2], batch_size=4)
    assert maps.shape == (2, 10, 40, 40) and np.isfinite(maps).all() and np.abs(maps).max() > 0


def test_keras_weight_list_interchange_and_masked_decode(tmp_path):
    """weights in keras get_weights() order travel through .npz (and, opt-in, the
    reference's pickle, train.py:287-289), trailing metric scalars are ignored; masked decode = decode of z
    with one unit set to its np.median (analysis.py:574-581)."""
    import pickle
    import numpy as np
    from oracle import vae_oracle as vo
    from vae_eeg_xai_amd.models import Encoder, Decoder, VAE
    L = 8
    a = VAE(Encoder(L), Decoder()); b = VAE(Encoder(L), Decoder()); c = VAE(Encoder(L), Decoder())
    ws = a.get_weights()
    names = [n for n, *_ in vo.param_spec(L, "vae")]
    assert len(ws) == len(names) == 62
    np.savez(tmp_path / "w.npz", *ws)
    b.load_keras_weight_list(tmp_path / "w.npz")
    with open(tmp_path / "w.pickle", "wb") as fp:
        pickle.dump(ws + [np.float32(3.0), np.float32(7.0)], fp)       # + metric total / count
    with pytest.raises(ValueError):
        c.load_keras_weight_list(tmp_path / "w.pickle")
    c.load_keras_weight_list(tmp_path / "w.pickle", allow_pickle=True)
    for m in (b, c):
        for u, v in zip(ws, m.get_weights()):
            assert np.array_equal(u, v)
    # the pre-round-2 flat order (all trainable, then all moving statistics) is still accepted, by its shapes
    specs = a._net.specs
    flat = [w for s, w in zip(specs, ws) if s.trainable] + [w for s, w in zip(specs, ws) if not s.trainable]
    assert [w.shape for w in flat] != [w.shape for w in ws]
    d = VAE(Encoder(L), Decoder())
    d.set_weights(flat)
    for u, v in zip(ws, d.get_weights()):
        assert np.array_equal(u, v)
    # keras order: encoder trainable (18), encoder moving statistics (6), decoder trainable (26), decoder's (12)
    assert names[18] == "enc.conv1.mean" and names[24] == "dec.dense.kernel" and names[50] == "dec.dense.mean"
    for N in (11, 12):       # odd and even N: np.median is the fp32 mean of the two middle elements
        z = np.random.default_rng(N).standard_normal((N, L)).astype(np.float32)
        zm = z.copy(); zm[:, 3] = np.median(z[:, 3])
        assert np.array_equal(np.asarray(a.decode_masked(z, 3)), np.asarray(a.decoder(zm))), N
    from vae_eeg_xai_amd.latent_stats import column_medians
    for N in (1, 2, 7, 1000, 4097):
        z = np.random.default_rng(N).standard_normal((N, L)).astype(np.float32)
        assert np.array_equal(column_medians(z).cpu().numpy(), np.median(z, axis=0)), N
    z = np.random.default_rng(0).standard_normal((11, L)).astype(np.float32)
    zv = z.copy(); zv[:, 0] = 0.25
    assert np.array_equal(np.asarray(a.decode_masked(z, 0, value=0.25)), np.asarray(a.decoder(zv)))


def test_device_array_indexes_on_the_device():
    """analysis.py indexes whole-set outputs per sample (analysis.py:46,581,590): same values as ndarray indexing."""
    import torch
    from vae_eeg_xai_amd.models import DeviceArray
    a = np.random.default_rng(0).standard_normal((9, 4, 3)).astype(np.float32)
    d = DeviceArray(torch.from_numpy(a).cuda())
    idx = np.array([3, 0, 8])
    mask = a[:, 0, 0] > 0
    for key in (2, -1, slice(1, 7, 2), (slice(None), 1), (3, 2, 1), idx, mask, [1, 2], (idx, 0), (Ellipsis, 2),
                np.int64(4), slice(None, None, -1)):
        got, ref = d[key], a[key]
        assert np.array_equal(got, ref) and np.shape(got) == np.shape(ref), key
    assert isinstance(d[1, 2, 0], np.float32)
    with pytest.raises(IndexError):
        d[9]

