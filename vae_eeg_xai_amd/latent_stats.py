"""Latent statistics on the device: the step right after the hot path in the
reference's ``analysis.py`` -- per-unit quantile thresholds of ``z_mean`` (``histogram_25_75``,
analysis.py:214-301), the TP/FN/FP/TN counts and the ROC-AUC ranking of latent units (``auc_roc``,
analysis.py:304-376) whose top-k indices select the units the gradient attribution targets
(analysis.py:725-726).  Same names, argument meaning and return structure as the reference (minus the
matplotlib figures).  Quantile selection and counting run in libveex.so (latent_stats.hip); only the
18-point AUC per unit is host arithmetic.  No CPU path: without the library / a GPU these raise."""
from __future__ import annotations

import ctypes as C
from typing import List, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import lib, check, ptr
from .dataio import _dev


def column_quantiles(z, q) -> torch.Tensor:
    """``np.quantile(z, q, axis=0)`` (NumPy 1.24 semantics for a float32 ``z`` and Python-float orders:
    float64 result) -> device float64 tensor of shape (len(q), L), or (L,) for a scalar ``q``."""
    t = _dev(z)
    if t.dim() != 2:
        raise ValueError("column_quantiles expects an (N, L) matrix")
    scalar = np.ndim(q) == 0
    qs = np.atleast_1d(np.asarray(q, dtype=np.float64))
    N, L = t.shape
    out = torch.empty((len(qs), L), dtype=torch.float64, device=t.device)
    for s in range(0, len(qs), 24):   # the kernel tracks at most 48 ranks (24 quantiles) per launch
        part = np.ascontiguousarray(qs[s:s + 24])
        nq = len(part)
        scratch = torch.empty(lib.veex_column_quantiles_scratch_bytes(nq, L), dtype=torch.uint8, device=t.device)
        check(lib.veex_column_quantiles(ptr(t), N, L, part.ctypes.data_as(C.POINTER(C.c_double)), nq,
                                        C.c_void_p(out[s:s + nq].data_ptr()), ptr(scratch), _lib.current_stream()),
              "column_quantiles")
    return out[0] if scalar else out


def column_medians(z) -> torch.Tensor:
    """``np.median(z[:, j])`` of every column of a float32 (N, L) matrix -> device float32 tensor (L,):
    the value the reference's latent masking writes into a unit (analysis.py:574-577)."""
    t = _dev(z)
    if t.dim() != 2:
        raise ValueError("column_medians expects an (N, L) matrix")
    N, L = t.shape
    out = torch.empty(L, dtype=torch.float32, device=t.device)
    scratch = torch.empty(lib.veex_column_quantiles_scratch_bytes(1, L), dtype=torch.uint8, device=t.device)
    check(lib.veex_column_medians(ptr(t), N, L, ptr(out), ptr(scratch), _lib.current_stream()), "column_medians")
    return out


def band_outside_counts(z, lo, hi) -> torch.Tensor:
    """counts[j, k] = #{i : z[i, j] <= lo[j, k] or z[i, j] >= hi[j, k]} (int32, device); lo/hi: (L, nb)."""
    t = _dev(z)
    N, L = t.shape
    def _thr(a):
        if isinstance(a, np.ndarray):
            a = np.ascontiguousarray(a)
        return torch.as_tensor(a, dtype=torch.float64, device=t.device).reshape(L, -1).contiguous()
    lo_t, hi_t = _thr(lo), _thr(hi)
    nb = lo_t.shape[1]
    counts = torch.empty((L, nb), dtype=torch.int32, device=t.device)
    check(lib.veex_band_outside_counts(ptr(t), N, L, nb, ptr(lo_t), ptr(hi_t), ptr(counts), _lib.current_stream()),
          "band_outside_counts")
    return counts


def quantile_matrix(z_mean) -> np.ndarray:
    """analysis.py:287-300: M[:, i] = quantile(z_mean, q) for q in arange(.05, 1.0, .05) without 0.5 -> (L, 18)."""
    start, end, step = 0.05, 0.95, 0.05
    Q = np.arange(start, end + step, step)
    qs = [q for q in Q if q != 0.5]
    M = column_quantiles(z_mean, qs).cpu().numpy()   # (18, L)
    return np.ascontiguousarray(M.T)


def histogram_25_75_counts(z_mean, z_mean_blink, z_mean_no_blink):
    """analysis.py:243-264: quartile thresholds of ALL z_mean, then per unit TP/FN over the blink rows and
    TN/FP over the no-blink rows.  Returns (quantile_25, quantile_75, TP, FN, FP, TN) as NumPy arrays (L,)."""
    q = column_quantiles(z_mean, [0.25, 0.75])
    lo, hi = q[0].reshape(-1, 1), q[1].reshape(-1, 1)
    zb, zn = _dev(z_mean_blink), _dev(z_mean_no_blink)
    tp = band_outside_counts(zb, lo, hi).cpu().numpy()[:, 0]
    fp = band_outside_counts(zn, lo, hi).cpu().numpy()[:, 0]
    return (q[0].cpu().numpy(), q[1].cpu().numpy(), tp, zb.shape[0] - tp, fp, zn.shape[0] - fp)


def _roc_auc(y_true: np.ndarray, y_score: np.ndarray) -> float:
    """sklearn.metrics.roc_auc_score for binary labels (analysis.py:339): Mann-Whitney U with mid-ranks."""
    order = np.argsort(y_score, kind="mergesort")
    s = y_score[order]
    ranks = np.empty(len(s), dtype=np.float64)
    i = 0
    while i < len(s):
        j = i
        while j + 1 < len(s) and s[j + 1] == s[i]:
            j += 1
        ranks[i:j + 1] = 0.5 * (i + j) + 1.0
        i = j + 1
    r = np.empty_like(ranks)
    r[order] = ranks
    n1 = float(np.sum(y_true == 1)); n0 = float(len(y_true) - n1)
    return float((np.sum(r[y_true == 1]) - n1 * (n1 + 1) / 2.0) / (n1 * n0))


def auc_roc(quantile_matrix_, z_mean_blink, z_mean_no_blink, subject=None) -> List[Tuple[int, float]]:
    """analysis.py:304-376 without the figure: for every unit j and every band (q, -q-1) of the quantile
    matrix, TPR = TP/(TP+FN) over the blink rows and FPR = FP/(FP+TN) over the no-blink rows; the unit's score
    is roc_auc_score(zeros(len(FPR)) ++ ones(len(TPR)), FPR ++ TPR); returns [(j, auc)] sorted by auc, descending."""
    M = np.asarray(quantile_matrix_, dtype=np.float64)
    L, nb2 = M.shape
    nb = nb2 // 2
    lo = np.ascontiguousarray(M[:, :nb])
    hi = np.ascontiguousarray(M[:, ::-1][:, :nb])     # column -q-1
    zb, zn = _dev(z_mean_blink), _dev(z_mean_no_blink)
    tp = band_outside_counts(zb, lo, hi).cpu().numpy().astype(np.float64)
    fp = band_outside_counts(zn, lo, hi).cpu().numpy().astype(np.float64)
    tpr, fpr = tp / zb.shape[0], fp / zn.shape[0]
    y = np.concatenate((np.zeros(nb), np.ones(nb)))
    auc_list = [(j, _roc_auc(y, np.concatenate((fpr[j], tpr[j])))) for j in range(L)]
    return sorted(auc_list, key=lambda x: x[1], reverse=True)
