"""A train.py-shaped driver (this package's own code, NOT the reference's script): the call sequence of
reference train.py:120-162 (refit), :172-237 (CustomGridSearchCV.fit) and analysis.py:667-672 (rebuild,
train_on_batch, load_weights), written against the names the reference imports -- ``from models import *``,
``keras.optimizers.legacy.Adam``, ``keras.src.callbacks.EarlyStopping``, ``np_config`` -- which resolve to
this package when ``compat/`` is first on sys.path.  The score is 1 / (mse + 1): the reference multiplies it
by a scaled SSIM from scikit-image, which is not installed here (out of scope).

    python examples/train_driver.py            # tiny synthetic run on cuda:0
"""
import os
import sys
from itertools import product

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "compat"))

from keras.optimizers.legacy import Adam                      # noqa: E402  (reference train.py:9)
from keras.src.callbacks import EarlyStopping                 # noqa: E402  (reference train.py:10)
from sklearn.model_selection import KFold, train_test_split   # noqa: E402  (reference train.py:14)
from tensorflow.python.ops.numpy_ops import np_config         # noqa: E402  (reference train.py:15)

from models import *                                          # noqa: E402,F401,F403  (reference train.py:18)

np_config.enable_numpy_behavior()                             # reference analysis.py:23


def score(x_true, x_pred):
    mse = float(np.mean((np.asarray(x_true, dtype=np.float64) - np.asarray(x_pred, dtype=np.float64)) ** 2))
    return 1.0 / (mse + 1.0), mse


def grid_search(x_train, latent_dimension, param_grid, n_splits=5, fit_seed=0):
    """reference train.py:172-237: every combination x KFold(n_splits): fresh model, fit with early stopping on
    the held-out fold, predict it, score; best = highest mean score."""
    keys = list(param_grid)
    results = []
    for combo in product(*(param_grid[k] for k in keys)):
        params = dict(zip(keys, combo))
        fold_scores = []
        for train_idx, val_idx in KFold(n_splits=n_splits).split(x_train):
            x_tr, x_va = x_train[train_idx], x_train[val_idx]
            encoder = Encoder(latent_dimension)
            decoder = Decoder()
            vae = VAE(encoder, decoder, params["epochs"], params["l_rate"], params["batch_size"], params["patience"])
            vae.compile(Adam(params["l_rate"]))
            early_stopping = EarlyStopping("val_loss", patience=params["patience"], verbose=0)
            vae.fit(x_tr, x_tr, params["batch_size"], params["epochs"], validation_data=(x_va, x_va),
                    callbacks=[early_stopping], verbose=0, seed=fit_seed)
            s, _ = score(x_va, vae.predict(x_va, verbose=0))
            fold_scores.append(s)
            tf.keras.backend.clear_session()              # reference train.py:207
        results.append((params, float(np.mean(fold_scores))))
    best = max(results, key=lambda r: r[1])
    return best[0], results


def refit(x_train, latent_dimension, best_params, fit_seed=0):
    """reference train.py:120-162: 80/20 split (random_state=42), fit the best combination once more."""
    x_tr, x_va = train_test_split(x_train, test_size=0.2, random_state=42)
    encoder = Encoder(latent_dimension)
    decoder = Decoder()
    vae = VAE(encoder, decoder, best_params["epochs"], best_params["l_rate"], best_params["batch_size"],
              best_params["patience"])
    vae.compile(Adam(best_params["l_rate"]))
    early_stopping = EarlyStopping("val_loss", patience=best_params["patience"], verbose=0)
    history = vae.fit(x_tr, x_tr, best_params["batch_size"], best_params["epochs"], validation_data=(x_va, x_va),
                      callbacks=[early_stopping], verbose=0, seed=fit_seed)
    return history, vae, (x_tr, x_va)


def reload(path, latent_dimension, l_rate, x_train):
    """reference analysis.py:667-672: new objects, one real optimizer step to create the variables, load_weights."""
    encoder = Encoder(latent_dimension)
    decoder = Decoder()
    vae = VAE(encoder, decoder)
    vae.compile(Adam(l_rate))
    vae.train_on_batch(x_train[:1], x_train[:1])
    vae.load_weights(path)
    return vae


def main(out_dir, n=96, latent_dimension=8):
    from vae_eeg_xai_amd.dataio import expand, normalize
    sys.path.insert(0, ROOT)
    from oracle.vae_oracle import synth_topomaps
    print(tf.config.list_physical_devices("GPU"))          # reference train.py:248
    x = np.asarray(normalize(expand(synth_topomaps(n, 3)[..., 0])).cpu())
    grid = {"epochs": [2], "l_rate": [1e-3, 1e-4], "batch_size": [16], "patience": [30]}
    best, results = grid_search(x, latent_dimension, grid, n_splits=2)
    history, vae, _ = refit(x, latent_dimension, best)
    path = os.path.join(out_dir, "vae_s01")
    vae.save_weights(path, save_format="tf")               # reference train.py:282
    vae2 = reload(path, latent_dimension, best["l_rate"], x)
    return best, results, history, vae, vae2, x


if __name__ == "__main__":
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        best, results, history, *_ = main(d)
    print("best", best, "results", results, "val_loss", history.history["val_loss"])
