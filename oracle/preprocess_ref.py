"""ORACLE (test infrastructure -- never imported by the product path).

numpy restatement of the reference's local contrast normalisation:

* lcn(..., mode="reflect")  <- reference CellTracker/preprocess.py:85-114  (lcn_cpu; scipy.ndimage.convolve, reflect)
* lcn(..., mode="constant") <- reference CellTracker/preprocess.py:136-167 (lcn_gpu; Keras Conv3D with a ones kernel,
                                 'same' zero padding, result divided by the window volume)
* normalize_image           <- reference CellTracker/preprocess.py:170-188 (_normalize_image: median subtract, clamp, lcn_gpu)

PARITY STATUS.  The reflect variant is PINNED against golden vectors produced by the reference's own lcn_cpu
(tests/golden/make_golden.py -> tests/golden/preprocess.npz).  The zero-padded variant differs only in the border
handling; in the reference it executes inside tensorflow==2.11 (float32 Conv3D, unknown summation order) -> "parity
unpinned" for the fp32 rounding, restated from the formula.
"""
from __future__ import annotations

import numpy as np


def box_sum(a: np.ndarray, size, mode: str) -> np.ndarray:
    """Sum over a centred window of odd `size` along x, y, z; borders 'reflect' (numpy 'symmetric', as
    scipy.ndimage 'reflect': d c b a | a b c d | d c b a) or 'constant' zeros."""
    out = np.asarray(a, dtype=np.float64)
    for ax, k in enumerate(size):
        if k == 1:
            continue
        h = k // 2
        pad = [(0, 0)] * 3
        pad[ax] = (h, h)
        p = np.pad(out, pad, mode="symmetric" if mode == "reflect" else "constant")
        c = np.cumsum(p, axis=ax)
        c = np.concatenate([np.zeros_like(np.take(c, [0], axis=ax)), c], axis=ax)
        n = out.shape[ax]
        out = np.take(c, np.arange(k, k + n), axis=ax) - np.take(c, np.arange(0, n), axis=ax)
    return out


def lcn(img3d: np.ndarray, noise_level: float, filter_size=(27, 27, 1), mode: str = "constant") -> np.ndarray:
    vol = float(filter_size[0] * filter_size[1] * filter_size[2])
    x = np.asarray(img3d, dtype=np.float64)
    avg = box_sum(x, filter_size, mode) / vol
    std = np.sqrt(box_sum(np.square(x - avg), filter_size, mode) / vol)
    return (x - avg) / (std + noise_level)


def normalize_image(image: np.ndarray, noise_level: float) -> np.ndarray:
    x = image - np.median(image)
    x = np.where(x < 0, 0, x)
    return lcn(x, noise_level, (27, 27, 1), mode="constant")
