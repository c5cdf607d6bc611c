"""The few Keras / TensorFlow names the reference's drivers use around the hot path
(train.py:9-10,137-142,193-197,207,248; analysis.py:23,630,670-672), so that a train.py-shaped
driver runs against this package without edits to its logic.  No TensorFlow is involved."""
from __future__ import annotations

import numpy as np


class Adam:
    """keras.optimizers.legacy.Adam(lr): only carries the hyper-parameters; the update itself is
    the fused HIP kernel veex_adam (TF epsilon placement)."""

    def __init__(self, learning_rate=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-7, **kwargs):
        if beta_1 != 0.9 or beta_2 != 0.999 or epsilon != 1e-7:
            raise NotImplementedError("only the Keras default beta_1/beta_2/epsilon are supported")
        self.learning_rate = float(learning_rate)
        self.lr = self.learning_rate


class EarlyStopping:
    """keras.callbacks.EarlyStopping(monitor, patience, verbose) with Keras defaults
    (min_delta=0, mode auto->min for *loss, restore_best_weights=False) [KERAS-SEM]."""

    def __init__(self, monitor="val_loss", patience=0, verbose=0, min_delta=0.0, **kwargs):
        self.monitor, self.patience, self.verbose, self.min_delta = monitor, int(patience), verbose, abs(min_delta)
        self.wait = 0
        self.best = np.inf
        self.stopped_epoch = 0
        self.stop_training = False

    def on_train_begin(self):
        self.wait, self.best, self.stopped_epoch, self.stop_training = 0, np.inf, 0, False

    def on_epoch_end(self, epoch, logs):
        cur = logs.get(self.monitor)
        if cur is None:
            return
        self.wait += 1
        if cur < self.best - self.min_delta:
            self.best = cur
            self.wait = 0
            return                      # Keras returns right after an improvement (also with patience=0)
        if self.wait >= self.patience and epoch > 0:
            self.stopped_epoch = epoch
            self.stop_training = True
            if self.verbose:
                print(f"Epoch {epoch + 1}: early stopping")


class History:
    def __init__(self):
        self.history = {}
        self.epoch = []

    def append(self, epoch, logs):
        self.epoch.append(epoch)
        for k, v in logs.items():
            self.history.setdefault(k, []).append(v)


class _Config:
    @staticmethod
    def list_physical_devices(kind=None):
        import torch
        if kind in (None, "GPU"):
            return [f"PhysicalDevice(name='/physical_device:GPU:{i}', device_type='GPU')"
                    for i in range(torch.cuda.device_count())]
        return []


class _Backend:
    @staticmethod
    def clear_session():
        import gc
        import torch
        gc.collect()
        if torch.cuda.is_available():
            torch.cuda.empty_cache()


class _KerasNS:
    backend = _Backend()


class _TFNamespace:
    """Stand-in for the ``tf`` name that ``from models import *`` leaks in the reference."""
    config = _Config()
    keras = _KerasNS()


tf = _TFNamespace()
