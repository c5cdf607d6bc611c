"""NumPy restatement of the latent statistics of the reference's analysis.py (TEST INFRASTRUCTURE ONLY; see
oracle/__init__.py).  PARITY UNPINNED by reference fixtures (the reference has none).  The counting loops are
the reference's own Python loops, vectorised; ``quantile`` spells out NumPy 1.24's np.quantile for a float32
array and Python-float orders (requirements.txt pins numpy 1.24.3: float32 neighbour difference, float64
weight and result), because NumPy 2 changed the dtype of that computation."""
from __future__ import annotations

import numpy as np


def quantile(z, q):
    """np.quantile(z, q, axis=0), method 'linear', as NumPy 1.24 (analysis.py:243-244,298)."""
    z = np.asarray(z, dtype=np.float32)
    n = z.shape[0]
    s = np.sort(z, axis=0)
    qs = np.atleast_1d(np.asarray(q, dtype=np.float64))
    out = np.empty((len(qs),) + z.shape[1:], dtype=np.float64)
    for i, qq in enumerate(qs):
        v = qq * (n - 1)
        lo = int(np.floor(v)); hi = min(lo + 1, n - 1); lo = min(lo, n - 1)
        t = v - np.floor(v)
        a, b = s[lo], s[hi]
        d = (b - a)                                   # float32
        r = a.astype(np.float64) + d.astype(np.float64) * t
        if t >= 0.5:
            r = b.astype(np.float64) - d.astype(np.float64) * (1 - t)
        out[i] = r
    return out[0] if np.ndim(q) == 0 else out


def histogram_25_75_counts(z_mean, z_blink, z_noblink):
    """analysis.py:243-264, loop for loop."""
    q25, q75 = quantile(z_mean, .25), quantile(z_mean, .75)
    L = z_mean.shape[1]
    tp = np.zeros(L, int); fn = np.zeros(L, int); fp = np.zeros(L, int); tn = np.zeros(L, int)
    for j in range(L):
        for v in z_blink[:, j]:
            if v <= q25[j] or v >= q75[j]:
                tp[j] += 1
            else:
                fn[j] += 1
        for v in z_noblink[:, j]:
            if q25[j] < v < q75[j]:
                tn[j] += 1
            else:
                fp[j] += 1
    return q25, q75, tp, fn, fp, tn


def quantile_matrix(z_mean):
    """analysis.py:287-300."""
    L = z_mean.shape[1]
    M = np.zeros((L, 18))
    Q = np.arange(0.05, 0.95 + 0.05, 0.05)
    i = 0
    for q in Q:
        if q != 0.5:
            M[:, i] = quantile(z_mean, q)
            i += 1
    return M


def auc_roc(qm, z_blink, z_noblink, roc_auc_score):
    """analysis.py:314-373 (counting loops vectorised; roc_auc_score passed in: sklearn's when importable)."""
    out = []
    for j in range(z_blink.shape[1]):
        FPR, TPR = [], []
        for q in range(qm.shape[1] // 2):
            lo, hi = qm[j, q], qm[j, -q - 1]
            tb = np.sum((z_blink[:, j] <= lo) | (z_blink[:, j] >= hi)); nb_ = z_blink.shape[0] - tb
            tnn = np.sum((lo < z_noblink[:, j]) & (z_noblink[:, j] < hi)); fpp = z_noblink.shape[0] - tnn
            FPR.append(fpp / (fpp + tnn)); TPR.append(tb / (tb + nb_))
        auc = roc_auc_score(np.concatenate((np.zeros(len(FPR)), np.ones(len(TPR)))), np.concatenate((FPR, TPR)))
        out.append((j, auc))
    return sorted(out, key=lambda x: x[1], reverse=True)
