"""Probability map -> labelled regions -> cell centres on the GPU.

Reference: ``Tracker._segment`` (CellTracker/tracker.py:636-648) = U-Net prob map -> ``_watershed`` (:671-684, skimage
marker watershed in watershed.py:16-108) -> ``scipy.ndimage.center_of_mass(regions > 0, regions, 1..n)`` ->
``_transform_layer_to_real``.

The skimage watershed has no runnable reference in this image and is not rebuilt; the region step here is
threshold + 3D connected components (touching cells are not split).  Label numbering (raster order, small regions
removed, renumbered 1..n) and the centre-of-mass call follow the reference, so the centres feed ``Tracker.match`` /
``TrackerLite`` exactly like ``seg/coords%06d.npy`` does (raw voxel coordinates, float64).
"""
from __future__ import annotations

import numpy as np

from . import _dev, _lib


def segment_centroids_device(prob, threshold: float = 0.5, connectivity: int = 1, min_size: int = 0, cap: int = 4096,
                             want_labels: bool = True):
    """prob: contiguous float32 cuda tensor [x, y, z] -> (labels int32 cuda | None, centres fp64 cuda [n, 3], sizes int32 cuda [n]).

    `cap` is the initial capacity of the centre table; it is doubled and the call repeated when the volume holds more regions."""
    t = _dev.torch(); L = _lib.lib()
    if prob.dim() != 3 or not prob.is_cuda or not prob.is_contiguous() or prob.dtype != t.float32:
        raise ValueError("expected a contiguous float32 cuda tensor (x, y, z)")
    if connectivity not in (1, 2, 3):
        raise ValueError("connectivity must be 1, 2 or 3")
    if min_size < 0 or cap <= 0:
        raise ValueError("min_size must be >= 0 and cap positive")
    dims = _lib.ivec(prob.shape)
    labels = _dev.empty(tuple(prob.shape), t.int32, prob.device) if want_labels else None
    n_dev = _dev.empty((1,), t.int32, prob.device)
    while True:
        centres = _dev.empty((cap, 3), t.float64, prob.device)
        sizes = _dev.empty((cap,), t.int32, prob.device)
        nbytes = L.ct_segment_workspace_bytes(dims, int(cap))
        if nbytes == 0:
            raise ValueError(f"volume {tuple(prob.shape)} is too large for int32 voxel indices")
        ws = _dev.workspace(nbytes, prob.device)
        _lib.check(L.ct_segment_centroids(prob.data_ptr(), dims, float(threshold), int(connectivity), int(min_size),
                                          int(cap), labels.data_ptr() if want_labels else None, centres.data_ptr(),
                                          sizes.data_ptr(), n_dev.data_ptr(), ws.data_ptr(), ws.numel(),
                                          _dev.stream(prob.device)), "ct_segment_centroids")
        n = int(n_dev.item())
        if n <= cap:
            return labels, centres[:n], sizes[:n]
        cap = max(2 * cap, n)


def segment_centroids(prob, threshold: float = 0.5, connectivity: int = 1, min_size: int = 0):
    """numpy (x, y, z) prob map -> (labels int32, centres float64 [n, 3], sizes int32 [n]).

    Raises ValueError("No cell was detected ...") like tracker.py:637-643 when nothing exceeds the threshold / survives."""
    t = _dev.torch()
    a = np.asarray(prob)
    if a.ndim == 5:                       # the reference passes unet3_prediction's [1, x, y, z, 1]
        a = a[0, :, :, :, 0]
    if a.ndim != 3:
        raise ValueError(f"expected a 3-D probability map (x, y, z), got shape {a.shape}")
    d = t.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
    labels, centres, sizes = segment_centroids_device(d, threshold, connectivity, min_size)
    if centres.shape[0] == 0:
        raise ValueError("No cell was detected! Try to reduce the min_size / noise_level.")
    return labels.cpu().numpy(), centres.cpu().numpy(), sizes.cpu().numpy()
