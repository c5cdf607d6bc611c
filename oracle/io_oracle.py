"""NumPy restatement of the steps either side of the hot path (TEST INFRASTRUCTURE ONLY; see
oracle/__init__.py).  PARITY UNPINNED by reference fixtures (the reference has none); these are the
reference's own NumPy expressions, so the restatement is the definition."""
from __future__ import annotations

import numpy as np


def expand(x):
    """train.py:95-96."""
    return np.expand_dims(x, -1).astype("float32")


def normalize(x):
    """train.py:99-100 (float32 in, float32 out)."""
    return (x - np.min(x)) / (np.max(x) - np.min(x))


def denormalize(x_normalized, x):
    """signal_reconstruction.py:92-95 (get_min_max = np.min / np.max of x)."""
    x_min, x_max = np.min(x), np.max(x)
    return (x_normalized * (x_max - x_min)) + x_min


def matrix_indexes_from_2d_positions(x_pos, y_pos, min_x, min_y, max_x, max_y, num_channels, yellow, map_length):
    """indian_functions.py:164-194, branch for branch."""
    x = ((x_pos - min_x) * (num_channels - 1 - 0)) / (max_x - min_x)
    y = ((y_pos - min_y) * (num_channels - 1 - 0)) / (max_y - min_y)
    ix, iy = round(x), round(y)
    yl = np.asarray(yellow).reshape(-1, 2).tolist()
    if [ix, iy] not in yl:
        return ix, iy
    if ([ix, iy + 1] not in yl) and ([ix, iy + 1] != [ix, map_length]):
        return ix, iy + 1
    if [ix, iy - 1] not in yl:
        return ix, iy - 1
    if ([ix - 1, iy] not in yl) and ([ix - 1, iy] != [-1, iy]):
        return ix - 1, iy
    return ix + 1, iy


def channel_pixel_indexes(pos2d, yellow, num_channels=32, map_length=40):
    """indian_functions.py:197-226 from the projected 2-D positions on (the projection itself needs MNE)."""
    min_x = max_x = min_y = max_y = 0
    for p in pos2d:
        min_x = min(min_x, p[0]); min_y = min(min_y, p[1]); max_x = max(max_x, p[0]); max_y = max(max_y, p[1])
    return [matrix_indexes_from_2d_positions(p[0], p[1], min_x, min_y, max_x, max_y, num_channels, yellow, map_length)
            for p in pos2d]


def retrieve_channel_values(interpolated_map, pixel_coordinates):
    """indian_functions.py:146-160 with onlyValues=True."""
    return [interpolated_map[c[0]][c[1]] for c in pixel_coordinates]
