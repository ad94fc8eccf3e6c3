#!/usr/bin/env python3
"""Golden vectors of the marker watershed from THE REFERENCE'S OWN CODE ON THE REAL scikit-image.

scikit-image is not importable by the image's main interpreter (/usr/bin/python3, 3.10), but the image also carries an Anaconda
tree whose interpreter has it:  /opt/conda/bin/python3.9  with scikit-image 0.18.3 (the generation the reference's
`peak_local_max(indices=False)` needs), scipy 1.7.1, numpy 1.26.4.  Run there (build container only):

    PYTHONDONTWRITEBYTECODE=1 /opt/conda/bin/python3.9 -W ignore tests/golden/make_watershed_golden.py

numpy's sort: scikit-image orders peak candidates with np.argsort(-intensities), an UNSTABLE sort, so the choice among exactly tied candidates
is whatever numpy's quicksort leaves.  Every numpy before 1.25 -- i.e. every numpy the reference's `tensorflow==2.11` (requirements.txt:4) can
run with -- uses its generic introsort; numpy >= 1.25 dispatches argsort to an AVX-512 network sort where the CPU has one (this box does), with
another order among equal keys.  The script therefore re-executes itself with NPY_DISABLE_CPU_FEATURES naming the AVX-512 groups, which puts
numpy 1.26.4 on the generic code path: the recorded tie choices are those of the reference's pinned environment (and of any machine without
AVX-512).  With the dispatch left on, 45 tied pairs of the benchmark stack are resolved the other way round (96 of 8.4 M voxels).

What runs: /root/reference/CellTracker/watershed.py (`watershed_2d`, `watershed_3d`) and `Tracker._watershed` of
/root/reference/CellTracker/tracker.py:671-684, unmodified, on the synthetic probability maps of tests/_ws_cases.py -- with scikit-image,
scipy, tifffile, h5py, matplotlib and sklearn REAL and only tensorflow / stardist / csbdeep (absent in that tree too, untouched by this
path) replaced by inert stubs.  Only inputs' hashes and the reference's outputs are written (tests/golden/watershed_skimage.npz); no
reference source travels.  For every case also the per-stage outputs of the four scikit-image primitives the oracle restates
(peak_local_max, label, watershed, find_boundaries) as the reference calls them, so that a difference can be located.
tests/test_watershed_pin.py holds the oracle (CPU) and the device path (GPU) to these files.
"""
import importlib
import importlib.util
import sys
import types
from pathlib import Path
from unittest.mock import MagicMock

import os  # noqa: E402

_NO_AVX512 = "AVX512F AVX512CD AVX512_SKX AVX512_CLX AVX512_CNL AVX512_ICL"
if os.environ.get("NPY_DISABLE_CPU_FEATURES") != _NO_AVX512:          # before numpy is imported
    os.execve(sys.executable, [sys.executable, "-W", "ignore", *sys.argv], dict(os.environ, NPY_DISABLE_CPU_FEATURES=_NO_AVX512, PYTHONDONTWRITEBYTECODE="1"))

sys.dont_write_bytecode = True
HERE = Path(__file__).resolve().parent
REPO = HERE.parent.parent
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / "tests"))
sys.path.insert(0, "/root/reference")

import numpy as np  # noqa: E402
import scipy  # noqa: E402
import scipy.ndimage as ndi  # noqa: E402
import skimage  # noqa: E402
from skimage.feature import peak_local_max  # noqa: E402
from skimage.morphology import label  # noqa: E402
from skimage.segmentation import find_boundaries, watershed  # noqa: E402

import _ws_cases as cases  # noqa: E402


def _stub_absent():
    names = ["tensorflow", "tensorflow.keras", "tensorflow.keras.layers", "tensorflow.keras.models", "tensorflow.keras.preprocessing",
             "tensorflow.keras.preprocessing.image", "tensorflow.keras.backend", "csbdeep", "csbdeep.utils", "csbdeep.utils.tf",
             "csbdeep.models", "stardist", "stardist.models", "stardist.utils", "stardist.nms", "stardist.matching", "stardist.models.base",
             "stardist.geometry", "stardist.rays3d"]
    for n in names:
        m = MagicMock(name=n); m.__path__ = []; m.__name__ = n
        sys.modules[n] = m
    sys.modules["tensorflow.keras"].Model = type("Model", (), {})
    sys.modules["tensorflow.keras.models"].Model = sys.modules["tensorflow.keras"].Model
    sys.modules["stardist.models"].StarDist3D = type("StarDist3D", (), {})
    sys.modules["csbdeep.utils.tf"].keras_import = lambda sub, *nm: MagicMock() if len(nm) <= 1 else tuple(MagicMock() for _ in nm)


def main():
    _stub_absent()
    import matplotlib
    matplotlib.use("Agg")
    ref_ws = importlib.import_module("CellTracker.watershed")
    ref_tracker = importlib.import_module("CellTracker.tracker")
    assert ref_ws.peak_local_max is peak_local_max and ref_ws.watershed is watershed, "the reference must be bound to the real scikit-image"
    make_stack = importlib.import_module("3deecelltracker_amd.synth").make_stack
    from numpy.core._multiarray_umath import __cpu_features__ as cpu
    assert not cpu["AVX512_SKX"] and not cpu["AVX512F"], "numpy must run its generic sort (see the module docstring)"
    out = {"versions": np.array([f"scikit-image {skimage.__version__}", f"scipy {scipy.__version__}", f"numpy {np.__version__}",
                                 f"python {sys.version.split()[0]}", "numpy sort: generic introsort (AVX-512 dispatch disabled)"])}
    names = []
    for name, (build, zr, ms) in cases.PIN_CASES.items():
        prob = build(make_stack)
        big = prob.size > 4_000_000
        # the reference's composite, unmodified: Tracker._watershed -> watershed_2d -> watershed_3d -> relabel_sequential
        trk = object.__new__(ref_tracker.Tracker)
        trk.z_siz, trk.z_xy_ratio, trk.min_size, trk.cell_num, trk.shrink = prob.shape[2], zr, ms, 0, (24, 24, 2)
        seg_auto = np.asarray(ref_tracker.Tracker._watershed(trk, prob[None, :, :, :, None], "min_size"))
        n = int(seg_auto.max())
        centres = np.asarray(ndi.center_of_mass(seg_auto > 0, seg_auto, range(1, n + 1)), dtype=np.float64).reshape(n, 3)   # tracker.py:646-647
        out[f"{name}_sha"] = np.array(cases.sha(prob)); out[f"{name}_shape"] = np.array(prob.shape)
        out[f"{name}_para"] = np.array([zr, ms, trk.min_size, trk.cell_num], dtype=np.float64)
        out[f"{name}_seg_auto"] = seg_auto.astype(np.int16); out[f"{name}_centres"] = centres
        assert n < 32767
        names.append(name)
        if big:
            # only the peak masks (sparse: a few KB): the choice among exactly tied candidates this run made (see oracle/watershed_ref.py)
            wo, _ = ref_ws.watershed_2d(prob, z_range=prob.shape[2], min_distance=7)
            peaks2d = np.zeros(prob.shape, bool)
            for z in range(prob.shape[2]):
                smooth = ndi.gaussian_filter(ndi.distance_transform_edt(prob[:, :, z] > 0.5, sampling=[1, 1]), 2, mode="constant")
                peaks2d[:, :, z] = peak_local_max(smooth, min_distance=7, indices=False)
            smooth3 = ndi.gaussian_filter(ndi.distance_transform_edt(wo, sampling=[1, 1, zr]), (2, 2, 0.3), mode="constant")
            out[f"{name}_peaks2d"] = np.packbits(peaks2d)
            out[f"{name}_peaks3d"] = np.packbits(peak_local_max(smooth3, min_distance=3, exclude_border=0, indices=False))
        if not big:
            # the cell_num method (watershed.py:86-88) on the same 2-D stage
            wo, bd = ref_ws.watershed_2d(prob, z_range=prob.shape[2], min_distance=7)
            _, clear_cn, ms2, cn2 = ref_ws.watershed_3d(wo, samplingrate=[1, 1, zr], method="cell_num", min_size=0, cell_num=max(trk.cell_num - 2, 1),
                                                        min_distance=3)
            out[f"{name}_wo2d"] = np.packbits(wo); out[f"{name}_bd2d"] = np.packbits(bd)
            out[f"{name}_cellnum"] = np.array([max(trk.cell_num - 2, 1), ms2, cn2]); out[f"{name}_clear_cellnum"] = clear_cn.astype(np.int16)
            # stage outputs of the primitives, called as watershed.py calls them
            peaks2d = np.zeros(prob.shape, bool); labels2d = np.zeros(prob.shape, np.int32)
            for z in range(prob.shape[2]):
                bn = prob[:, :, z] > 0.5
                smooth = ndi.gaussian_filter(ndi.distance_transform_edt(bn, sampling=[1, 1]), 2, mode="constant")
                pk = peak_local_max(smooth, min_distance=7, indices=False)
                peaks2d[:, :, z] = pk; labels2d[:, :, z] = watershed(-smooth, label(pk), mask=bn)
                assert np.array_equal(find_boundaries(labels2d[:, :, z], connectivity=2, mode="outer", background=0), bd[:, :, z])
            smooth3 = ndi.gaussian_filter(ndi.distance_transform_edt(wo, sampling=[1, 1, zr]), (2, 2, 0.3), mode="constant")
            pk3 = peak_local_max(smooth3, min_distance=3, exclude_border=0, indices=False)
            out[f"{name}_peaks2d"] = np.packbits(peaks2d); out[f"{name}_labels2d"] = labels2d.astype(np.int16)
            out[f"{name}_peaks3d"] = np.packbits(pk3); out[f"{name}_labels3d"] = watershed(-smooth3, label(pk3), mask=wo).astype(np.int16)
            out[f"{name}_smooth3_sha"] = np.array(cases.sha(smooth3))          # scipy 1.7.1's floats: informative only
        print(f"{name}: {prob.shape} z ratio {zr} min_size {ms} -> {n} cells (cell_num {trk.cell_num})")
    out["names"] = np.array(names)
    np.savez_compressed(HERE / "watershed_skimage.npz", **out)
    leftovers = [p for p in Path("/root/reference").rglob("__pycache__")]
    assert not leftovers, leftovers
    print("written", HERE / "watershed_skimage.npz", list(out["versions"]))


if __name__ == "__main__":
    main()
