"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the prob-map -> centres step.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

What is restated and how it is pinned:
  * centres: the reference's own call, tracker.py:646-647
        ndm.center_of_mass(segmentation > 0, segmentation, range(1, segmentation.max() + 1))
    scipy IS installed here, so this part runs the very function the reference runs -- pinned.
  * relabel: skimage.segmentation.relabel_sequential (tracker.py:680) -- order-preserving renumbering 1..n; restated.
  * small-object removal: skimage.morphology.remove_small_objects (watershed.py:96): drop labels whose voxel count is
    < min_size; restated from its documented behaviour.
  * regions: the reference uses a skimage marker watershed (watershed.py:16-108).  skimage is absent from this image, so
    there is no runnable reference: PARITY UNPINNED for that step, and it is not restated.  The region step here is
    scipy.ndimage.label on (prob > threshold) -- the connected-components variant SURVEY 8f#2 names.
"""
from __future__ import annotations

import numpy as np
import scipy.ndimage as ndm


def remove_small_objects(labels: np.ndarray, min_size: int) -> np.ndarray:
    sizes = np.bincount(labels.ravel())
    small = sizes < min_size
    small[0] = False
    out = labels.copy()
    out[small[labels]] = 0
    return out


def relabel_sequential(labels: np.ndarray) -> np.ndarray:
    present = np.unique(labels)
    present = present[present > 0]
    lut = np.zeros(int(labels.max()) + 1, dtype=np.int32)
    lut[present] = np.arange(1, present.size + 1, dtype=np.int32)
    return lut[labels]


def segment_centroids(prob: np.ndarray, threshold: float = 0.5, connectivity: int = 1, min_size: int = 0):
    """-> (labels int32 [x, y, z], centres float64 [n, 3], sizes int64 [n])."""
    prob = np.asarray(prob, dtype=np.float32)
    structure = ndm.generate_binary_structure(3, connectivity)
    labels, _ = ndm.label(prob > np.float32(threshold), structure=structure)
    labels = relabel_sequential(remove_small_objects(labels.astype(np.int32), min_size))
    n = int(labels.max())
    if n == 0:
        return labels, np.zeros((0, 3)), np.zeros(0, dtype=np.int64)
    centres = np.asarray(ndm.center_of_mass(labels > 0, labels, range(1, n + 1)), dtype=np.float64)
    sizes = np.bincount(labels.ravel())[1:]
    return labels, centres, sizes
