import glob
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
RTOL = 1e-5     # north_star: fp32 forward / loss / gradients within 1e-5 relative (normwise, max-abs scaled)


def rel_err(got, ref):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    return float(np.abs(got - ref).max() / (np.abs(ref).max() + 1e-30))


def golden_cases():
    out = []
    for path in sorted(glob.glob(os.path.join(GOLDEN_DIR, "*.npz"))):
        name = os.path.basename(path)[:-4]
        if "__" in name:          # alternate-[KERAS-SEM] siblings (make_golden.ALT_SEMS): alt_golden_cases()
            continue
        variant, L, B, s = name.split("_")
        out.append((path, variant, int(L[1:]), int(B[1:]), int(s[1:])))
    return out


ALT_SEMS = {"bn-batch": {"bn": "batch"}, "bce-probs": {"bce": "probs"}, "bn-batch+bce-probs": {"bn": "batch", "bce": "probs"}}


def alt_golden_cases():
    """(path, variant, L, B, seed, sem) of the fixtures made under an alternate Keras-semantics reading."""
    out = []
    for path, variant, L, B, seed in golden_cases():
        for tag, sem in ALT_SEMS.items():
            alt = path[:-4] + "__" + tag + ".npz"
            if os.path.exists(alt):
                out.append((alt, variant, L, B, seed, sem))
    return out


LAYERS = ["enc.conv1", "enc.conv2", "enc.conv3", "enc.dense", "dec.dense"] + [f"dec.deconv{i}" for i in range(1, 6)]


def device_masks(net, B, cache, layers=LAYERS, kink_tol=2e-5):
    """ReLU decisions of the device.  Every disagreement with the float64 oracle must sit at a
    pre-activation within fp32 rounding of zero (an ambiguous sub-gradient), otherwise it is a bug."""
    masks = {}
    for ln in layers:
        ref_pre = cache[ln + ".pre"]
        got = net.ws_region(B, "act." + ln).cpu().numpy().reshape(ref_pre.shape)
        m = got > 0
        dis = m != (ref_pre > 0)
        if dis.any():
            worst = np.abs(ref_pre[dis]).max()
            assert worst < kink_tol * max(1.0, np.abs(ref_pre).max()), (ln, int(dis.sum()), worst)
        masks[ln] = m
    return masks
