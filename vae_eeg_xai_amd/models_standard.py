"""Drop-in counterpart of the reference's ``models_standard.py`` (Autoencoder, EncoderStandard,
DecoderStandard; reference models_standard.py:10-175): same convolutional trunk and decoder as the
VAE, bottleneck Dense(6400 -> L, relu), loss = BCE term only."""
from __future__ import annotations

from .models import VAE, Encoder, Decoder, DeviceArray, _Mean
from .keras_compat import Adam, EarlyStopping, History, tf  # noqa: F401

keras = tf.keras


class EncoderStandard(Encoder):
    """EncoderStandard(latent_dimension)(x) -> bottleneck (B, L)  (models_standard.py:81-116)."""

    _variant = "ae"

    def __call__(self, inputs, training=None, mask=None):
        z, _, _ = self._net.encode(inputs)
        return DeviceArray(z)

    call = __call__


class DecoderStandard(Decoder):
    """DecoderStandard()(bottleneck) -> (B,40,40,1)  (models_standard.py:119-175)."""

    _variant = "ae"


class Autoencoder(VAE):
    """Autoencoder(encoder, decoder, epochs, l_rate, batch_size, patience)  (models_standard.py:10-78).
    One metric, ``loss`` (the reconstruction term)."""

    _metric_names = ("loss",)
    _variant = "ae"

    def __init__(self, encoder, decoder, epochs=None, l_rate=None, batch_size=None, patience=None, **kwargs):
        super().__init__(encoder, decoder, epochs, l_rate, batch_size, patience, **kwargs)
        self.loss_tracker = _Mean("loss", self, 0)

    @property
    def metrics(self):
        return [self.loss_tracker]


__all__ = ["Autoencoder", "EncoderStandard", "DecoderStandard", "Adam", "EarlyStopping", "History", "tf", "keras"]
