"""vae_eeg_xai_amd -- MI355X (gfx950) implementation of the VAE / autoencoder hot path of
alessandroquirile/VAE-EEG-XAI behind the reference's models.py / models_standard.py API.

Importing the package does not touch the GPU; the HIP library (libveex.so) is loaded on first use
of ``vae_eeg_xai_amd._lib`` and there is no CPU fallback.
"""
__version__ = "0.1.0"
