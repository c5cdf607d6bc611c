"""The steps either side of the hot path, on the device:
``expand`` / ``normalize`` (reference train.py:95-100) in front of the encoder, ``denormalize``
(signal_reconstruction.py:92-95) and the 32-electrode read-out (indian_functions.py:146-226)
behind the decoder / the attribution maps.  Arithmetic runs in libveex.so (dataio.hip) and is
bit-identical to the reference's NumPy float32 expressions; the pixel table of the electrodes is
small integer host logic and is restated here from the reference's if-chain."""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import lib, check, ptr
from .net import _as_device_f32


def expand(x):
    """train.py:95-96: add the channel axis, cast to float32 (host; no arithmetic)."""
    return np.expand_dims(np.asarray(x), -1).astype("float32")


def _dev(x, device=None) -> torch.Tensor:
    if not torch.cuda.is_available():
        raise _lib.VeexError("no GPU visible: vae_eeg_xai_amd has no CPU path")
    return _as_device_f32(x, torch.device(device or f"cuda:{torch.cuda.current_device()}"))


def minmax(x) -> torch.Tensor:
    """(min, max) over the whole array as a 2-float device tensor."""
    t = _dev(x)
    n = t.numel()
    part = torch.empty(2 * lib.veex_minmax_partial_blocks(n), dtype=torch.float32, device=t.device)
    out = torch.empty(2, dtype=torch.float32, device=t.device)
    check(lib.veex_minmax(ptr(t), n, ptr(part), ptr(out), _lib.current_stream()), "minmax")
    return out


def normalize(x) -> torch.Tensor:
    """train.py:99-100: (x - min(x)) / (max(x) - min(x)) with the extrema of the ENTIRE array."""
    t = _dev(x)
    mm = minmax(t)
    out = torch.empty_like(t)
    check(lib.veex_normalize(ptr(t), ptr(mm), ptr(out), t.numel(), _lib.current_stream()), "normalize")
    return out


def _extrema(x, minmax_pair) -> torch.Tensor:
    """(min, max) device pair: ``minmax_pair`` when the caller computed it before (``minmax(x)``), else of ``x``."""
    if minmax_pair is not None:
        mm = _dev(minmax_pair).reshape(-1)
        if mm.numel() != 2:
            raise ValueError("minmax= expects the 2-element (min, max) pair")
        return mm
    if x is None:
        raise ValueError("give the array the extrema come from, or minmax=")
    return minmax(x)


def denormalize(x_normalized, x=None, *, minmax=None) -> torch.Tensor:
    """signal_reconstruction.py:92-95: x_normalized * (max(x) - min(x)) + min(x).  ``minmax=`` takes extrema
    computed earlier (``dataio.minmax(x)``) instead of reducing ``x`` again."""
    tn = _dev(x_normalized)
    mm = _extrema(x, minmax)
    out = torch.empty_like(tn)
    check(lib.veex_denormalize(ptr(tn), ptr(mm), ptr(out), tn.numel(), _lib.current_stream()), "denormalize")
    return out


def electrode_pixel_table(pos2d, zero_region=(), n_channels: int = 32, map_length: int = 40) -> np.ndarray:
    """(row, col) of the pixel each electrode is read from: indian_functions.py:197-226 with the
    index arithmetic and the nudge-off-the-zero-region chain of :164-194.  ``pos2d``: (E,2) projected
    electrode positions (azim_proj of the montage, which needs MNE and is therefore an input);
    ``zero_region``: (K,2) integer pixels where the map is exactly 0 (np.argwhere(map == 0.)).
    Like the reference, the rescale uses ``n_channels - 1`` (not the map length) and min/max start at 0."""
    pos2d = np.asarray(pos2d, dtype=np.float64)
    zero = {(int(a), int(b)) for a, b in np.asarray(zero_region, dtype=np.int64).reshape(-1, 2)}
    min_x = min(0.0, float(pos2d[:, 0].min())); max_x = max(0.0, float(pos2d[:, 0].max()))
    min_y = min(0.0, float(pos2d[:, 1].min())); max_y = max(0.0, float(pos2d[:, 1].max()))
    out = np.empty((pos2d.shape[0], 2), dtype=np.int64)
    for e, (xp, yp) in enumerate(pos2d):
        ix = round(((xp - min_x) * (n_channels - 1)) / (max_x - min_x))
        iy = round(((yp - min_y) * (n_channels - 1)) / (max_y - min_y))
        if (ix, iy) not in zero:
            pass
        elif (ix, iy + 1) not in zero and iy + 1 != map_length:
            iy = iy + 1
        elif (ix, iy - 1) not in zero:
            iy = iy - 1
        elif (ix - 1, iy) not in zero and ix - 1 != -1:
            ix = ix - 1
        else:
            ix = ix + 1
        out[e] = (ix, iy)
    return out


def electrode_readout(maps, pixel_table, denormalize_with=None, map_shape: Tuple[int, int] = (40, 40), *,
                      minmax=None) -> torch.Tensor:
    """indian_functions.py:146-160 for a whole batch: out[b, e] = maps[b][row_e][col_e]
    (optionally denormalized with the extrema of the array ``denormalize_with``, or with a (min, max)
    pair computed earlier passed as ``minmax=``).  ``maps``: (B,H,W[,1])."""
    t = _dev(maps)
    H, W = map_shape
    if t.numel() % (H * W) != 0:
        raise ValueError(f"maps of shape {tuple(t.shape)} are not {H}x{W} images")
    B = t.numel() // (H * W)
    tab = np.asarray(pixel_table, dtype=np.int64).reshape(-1, 2)
    if (tab[:, 0] < 0).any() or (tab[:, 0] >= H).any() or (tab[:, 1] < 0).any() or (tab[:, 1] >= W).any():
        # stricter than the reference: NumPy would wrap a negative index to the other side of the map
        raise IndexError("electrode pixel outside the map")
    pix = torch.from_numpy((tab[:, 0] * W + tab[:, 1]).astype(np.int32)).to(t.device)
    mm = None
    if denormalize_with is not None or minmax is not None:
        mm = _extrema(denormalize_with, minmax)
    out = torch.empty(B, tab.shape[0], dtype=torch.float32, device=t.device)
    check(lib.veex_electrode_readout(ptr(t), ptr(pix), ptr(mm), ptr(out), B, H * W, tab.shape[0],
                                     _lib.current_stream()), "electrode_readout")
    return out
