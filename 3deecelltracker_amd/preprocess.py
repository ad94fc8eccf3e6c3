"""Host-side mirror of the normalisation functions of the reference's ``CellTracker/preprocess.py``
(SURVEY 8f next-row #1 -- the step that precedes the U-Net every frame):

    _normalize_image(image, noise_level)            reference preprocess.py:170-188
    lcn_gpu(img3d, noise_level=5, filter_size)      reference preprocess.py:136-167   (zero padding)
    lcn_cpu(img3d, noise_level, filter_size)        reference preprocess.py:85-114    (scipy 'reflect' borders)

numpy (x, y, z) in / numpy out; `normalize_image_device` keeps everything on the GPU (uint16 or float32 volume ->
float32 normalised volume, the median never visits the host).  Results are float32 (the reference's arithmetic
runs in Keras float32 and is then up-cast).
"""
from __future__ import annotations

import numpy as np

from . import _dev, _lib


def normalize_image_device(vol, noise_level: float, filter_size=(27, 27, 1), mode: int = 0, subtract_median: bool = True):
    """vol: torch cuda tensor [x, y, z], dtype uint16 (as int16 storage is NOT accepted), or float32."""
    t = _dev.torch(); L = _lib.lib()
    if vol.dim() != 3 or not vol.is_cuda or not vol.is_contiguous():
        raise ValueError("expected a contiguous 3-D cuda tensor (x, y, z)")
    if vol.dtype == t.uint16:
        dtype = 0
    elif vol.dtype == t.float32:
        dtype = 1
    else:
        raise TypeError(f"unsupported dtype {vol.dtype}: use uint16 or float32")
    if any(int(f) <= 0 or int(f) % 2 == 0 for f in filter_size):
        raise ValueError("filter sizes must be odd and positive")
    out = _dev.empty(tuple(vol.shape), t.float32, vol.device)
    ws = _dev.workspace(L.ct_normalize_workspace_bytes(_lib.ivec(vol.shape)), vol.device)
    _lib.check(L.ct_normalize_image(vol.data_ptr(), dtype, _lib.ivec(vol.shape), float(noise_level), _lib.ivec(filter_size),
                                    int(mode), int(bool(subtract_median)), out.data_ptr(), ws.data_ptr(), ws.numel(),
                                    _dev.stream(vol.device)), "ct_normalize_image")
    return out


def _to_device(img3d):
    t = _dev.torch()
    a = np.asarray(img3d)
    if a.ndim != 3:
        raise ValueError(f"expected a 3-D image (x, y, z), got shape {a.shape}")
    if a.dtype == np.uint16:
        return t.from_numpy(np.ascontiguousarray(a)).cuda()
    return t.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def lcn_gpu(img3d, noise_level=5, filter_size=(27, 27, 1)):
    return normalize_image_device(_to_device(img3d), noise_level, filter_size, mode=0, subtract_median=False).cpu().numpy()


def lcn_cpu(img3d, noise_level, filter_size=(27, 27, 1)):
    return normalize_image_device(_to_device(img3d), noise_level, filter_size, mode=1, subtract_median=False).cpu().numpy()


def _normalize_image(image, noise_level):
    return normalize_image_device(_to_device(image), noise_level, (27, 27, 1), mode=0, subtract_median=True).cpu().numpy()


def median_device(vol):
    """np.median of a uint16 / float32 cuda tensor -> python float (radix select on the device)."""
    t = _dev.torch(); L = _lib.lib()
    v = vol.contiguous().view(-1)
    dtype = 0 if v.dtype == t.uint16 else 1
    if dtype == 1 and v.dtype != t.float32:
        raise TypeError(f"unsupported dtype {v.dtype}")
    out = _dev.empty((1,), t.float64, v.device)
    ws = _dev.workspace(40960, v.device)                     # >= 4096; 16 histogram tables fit in 33 KB
    _lib.check(L.ct_median(v.data_ptr(), dtype, v.numel(), out.data_ptr(), ws.data_ptr(), ws.numel(), _dev.stream(v.device)), "ct_median")
    return float(out.item())
