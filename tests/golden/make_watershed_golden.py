#!/usr/bin/env python3
"""ONE-RUN KIT (cannot run in the build container: scikit-image is not installed and not installable there).

On any machine with scikit-image 0.16-0.19 (what the reference's `peak_local_max(indices=False)` needs), numpy and scipy:

    python tests/golden/make_watershed_golden.py            # writes tests/golden/watershed_skimage.npz

It records scikit-image's OWN outputs for the four primitives the oracle restates (oracle/watershed_ref.py) on the probability maps of
tests/golden/watershed.npz, and the full watershed_2d / watershed_3d chain built from them exactly as CellTracker/watershed.py:16-108
does.  With the file present, tests/test_watershed_pin.py holds the restatements -- and through them the device path -- to scikit-image
itself, and the "parity unpinned" note of the oracle can go.  Untested here by necessity.
"""
from pathlib import Path

import numpy as np
import scipy.ndimage as ndi

HERE = Path(__file__).resolve().parent


def main():
    import skimage
    from skimage.feature import peak_local_max
    from skimage.morphology import label, remove_small_objects
    from skimage.segmentation import find_boundaries, relabel_sequential, watershed
    g = np.load(HERE / "watershed.npz")
    out = {"skimage_version": np.array(skimage.__version__)}
    for ci in range(3):
        prob = g[f"ws_prob_{ci}"]
        zr, ms = float(g[f"ws_para_{ci}"][0]), int(g[f"ws_para_{ci}"][1])
        boundary = np.zeros(prob.shape, dtype=bool)
        peaks2d = np.zeros(prob.shape, dtype=bool); labels2d = np.zeros(prob.shape, dtype=np.int32)
        for z in range(prob.shape[2]):
            bn = prob[:, :, z] > 0.5
            dist = ndi.distance_transform_edt(bn, sampling=[1, 1])
            smooth = ndi.gaussian_filter(dist, 2, mode="constant")
            pk = peak_local_max(smooth, min_distance=7, indices=False)
            lab = watershed(-smooth, label(pk), mask=bn)
            peaks2d[:, :, z] = pk; labels2d[:, :, z] = lab
            boundary[:, :, z] = find_boundaries(lab, connectivity=2, mode="outer", background=0)
        wo = prob > 0.5
        wo[boundary] = False
        dist = ndi.distance_transform_edt(wo, sampling=[1, 1, zr])
        smooth = ndi.gaussian_filter(dist, (2, 2, 0.3), mode="constant")
        pk3 = peak_local_max(smooth, min_distance=3, exclude_border=0, indices=False)
        lab3 = watershed(-smooth, label(pk3), mask=wo)
        clear = remove_small_objects(lab3, min_size=ms, connectivity=3)
        seg, _, _ = relabel_sequential(clear)
        out[f"peaks2d_{ci}"] = np.packbits(peaks2d); out[f"labels2d_{ci}"] = labels2d.astype(np.int16)
        out[f"boundary2d_{ci}"] = np.packbits(boundary)
        out[f"peaks3d_{ci}"] = np.packbits(pk3); out[f"labels3d_{ci}"] = lab3.astype(np.int16)
        out[f"seg_auto_{ci}"] = np.asarray(seg).astype(np.int16)
    np.savez_compressed(HERE / "watershed_skimage.npz", **out)
    print("written", HERE / "watershed_skimage.npz", "with scikit-image", skimage.__version__)


if __name__ == "__main__":
    main()
