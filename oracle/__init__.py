"""CPU oracle for the VAE hot path -- TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it,
and there only as the checker.  The product path (``vae_eeg_xai_amd``) never
imports this package and fails loudly when its HIP library is missing.

PARITY UNPINNED: the reference (alessandroquirile/VAE-EEG-XAI) ships no tests,
fixtures or golden vectors, and its arithmetic lives in TensorFlow 2.13 /
Keras 2.13 (requirements.txt:26-27), which are not installed here and cannot be.
The oracle is therefore a restatement of the Keras semantics implied by
models.py:10-22,148-207,211-312 and models_standard.py:10-175 (SURVEY.md
Appendix A), cross-checked between two independent formulations
(``vae_oracle`` = NumPy float64 scatter/gather loops with a hand-derived
backward; ``torch_ref`` = torch CPU conv2d/conv_transpose2d + autograd) and by
central finite differences.
"""
