"""Generates tests/golden/*.npz from the float64 oracle (oracle/vae_oracle.py).

The reference cannot be imported here (TensorFlow/Keras absent), so these vectors pin
the build's own restatement -- "parity unpinned" with respect to a real TF run.  Weights are not
stored (9 MB): they are regenerated from the seed by oracle.vae_oracle.init_params.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import vae_oracle as vo  # noqa: E402

CASES = [("vae", 28, 2, 11), ("ae", 28, 2, 12), ("vae", 8, 3, 13)]


# the alternate [KERAS-SEM] readings (oracle/vae_oracle.py DEFAULT_SEM): one small fixture each, same inputs, so that
# compare_fixture.py can say WHICH reading a real TF run follows
ALT_SEMS = {"bn-batch": {"bn": "batch"}, "bce-probs": {"bce": "probs"}, "bn-batch+bce-probs": {"bn": "batch", "bce": "probs"}}


def build(variant, L, B, seed, sem=None):
    p = vo.init_params(L, variant, seed=seed, jitter=0.1)
    x = vo.synth_topomaps(B, seed + 100)
    eps = np.random.default_rng(seed + 200).standard_normal((B, L)).astype(np.float32) if variant == "vae" else None
    c = vo.forward(p, x, eps, variant, sem)
    g = vo.backward(p, c, variant)
    out = {"x": x, "total": c["total"], "rec_loss": c["rec_loss"], "kl_loss": c["kl_loss"],
           "logits": c["logits"].astype(np.float64), "recon": c["recon"], "z": c["z"],
           "enc_dense_a": c["enc.dense.a"]}
    if variant == "vae":
        out.update(eps=eps, z_mean=c["z_mean"], z_log_var=c["z_log_var"])
    names = vo.trainable_names(L, variant)
    for k in names:
        out["gsum." + k] = g[k].sum()
        out["gabs." + k] = np.abs(g[k]).sum()
        out["ghead." + k] = g[k].ravel()[:16].copy()
    # parameters after 1 and 3 Adam steps (lr 1e-3 so that the update is visible in fp32)
    m = {k: np.zeros_like(p[k]) for k in names}
    v = {k: np.zeros_like(p[k]) for k in names}
    pp = {k: a.copy() for k, a in p.items()}
    for t in (1, 2, 3):
        tot, rec, kl = vo.train_step(pp, m, v, t, 1e-3, x, eps, variant, sem)
        out[f"loss_step{t}"] = tot
        if t in (1, 3):
            for k in ("enc.conv2.kernel", "dec.deconv4.kernel", "dec.dense.gamma", "enc.conv1.bias"):
                out[f"p{t}.{k}"] = pp[k].ravel()[:32].copy()
    if sem is not None:                 # alternate reading: the scalars and gradient digests only
        return {k: v for k, v in out.items() if k.split(".")[0] in ("total", "rec_loss", "kl_loss", "gsum", "gabs", "ghead")
                or k.startswith("loss_step") or k in ("recon", "z_mean", "z_log_var")}
    units = [0, 5, L - 1]
    out["units"] = np.array(units)
    out["saliency"] = vo.attribution(p, x, units, variant)[..., 0]
    out["gxi"] = vo.attribution(p, x, units, variant, grad_x_input=True)[..., 0]
    return out


if __name__ == "__main__":
    for variant, L, B, seed in CASES:
        path = os.path.join(HERE, f"{variant}_L{L}_B{B}_s{seed}.npz")
        np.savez_compressed(path, **build(variant, L, B, seed))
        print(path, os.path.getsize(path))
        for tag, sem in ALT_SEMS.items():
            apath = os.path.join(HERE, f"{variant}_L{L}_B{B}_s{seed}__{tag}.npz")
            np.savez_compressed(apath, **build(variant, L, B, seed, sem))
            print(apath, os.path.getsize(apath))
