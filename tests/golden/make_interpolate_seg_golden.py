#!/usr/bin/env python3
"""Volume-1 set-up of the legacy Tracker from THE REFERENCE ITSELF with the real scikit-image / tifffile (second interpreter):
load_manual_seg (tracker.py:908-919: TIFF layers of a manually corrected segmentation, relabel_sequential) and interpolate_seg
(:1046-1085: z-interpolation, per-cell skimage.filters.gaussian smoothing, skimage.measure.label re-labelling, centres) on a small synthetic
label volume whose smoothed cells do not overlap.

    PYTHONDONTWRITEBYTECODE=1 /opt/conda/bin/python3.9 -W ignore tests/golden/make_interpolate_seg_golden.py

Writes tests/golden/manual_vol1/*.tif (the input layers, written by tifffile, < 3 KB) and tests/golden/interpolate_seg.npz (the reference's
results).  Only tensorflow / stardist / csbdeep are stubbed (absent, untouched by this path)."""
import contextlib
import importlib
import io
import sys
import tempfile
from pathlib import Path
from unittest.mock import MagicMock

sys.dont_write_bytecode = True
HERE = Path(__file__).resolve().parent
sys.path.insert(0, "/root/reference")
import numpy as np  # noqa: E402
import tifffile  # noqa: E402


def label_volume():
    shape = (40, 44, 6)
    lab = np.zeros(shape, dtype=np.uint8)
    gx, gy, gz = np.meshgrid(*(np.arange(s) for s in shape), indexing="ij")
    # labels 7, 3, 12 (not sequential: relabel_sequential has something to do), drawn out of raster order
    for l, c in ((7, (10, 10, 2)), (3, (28, 12, 3)), (12, (18, 32, 2.5))):
        lab[((gx - c[0]) / 4.0) ** 2 + ((gy - c[1]) / 4.0) ** 2 + ((gz - c[2]) / 1.2) ** 2 <= 1.0] = l
    return lab


def main():
    for n in ["tensorflow", "tensorflow.keras", "tensorflow.keras.layers", "tensorflow.keras.models", "tensorflow.keras.preprocessing",
              "tensorflow.keras.preprocessing.image", "tensorflow.keras.backend", "csbdeep", "csbdeep.utils", "csbdeep.utils.tf", "csbdeep.models",
              "stardist", "stardist.models", "stardist.utils", "stardist.nms", "stardist.matching", "stardist.models.base", "stardist.geometry",
              "stardist.rays3d"]:
        m = MagicMock(name=n); m.__path__ = []; m.__name__ = n; sys.modules[n] = m
    sys.modules["tensorflow.keras"].Model = type("Model", (), {})
    sys.modules["tensorflow.keras.models"].Model = sys.modules["tensorflow.keras"].Model
    sys.modules["stardist.models"].StarDist3D = type("StarDist3D", (), {})
    sys.modules["csbdeep.utils.tf"].keras_import = lambda sub, *nm: MagicMock() if len(nm) <= 1 else tuple(MagicMock() for _ in nm)
    import matplotlib
    matplotlib.use("Agg")
    ref_tracker = importlib.import_module("CellTracker.tracker")
    lab = label_volume()
    d = HERE / "manual_vol1"
    d.mkdir(exist_ok=True)
    for z in range(lab.shape[2]):
        tifffile.imwrite(d / ("manual_vol1_z%04i.tif" % (z + 1)), lab[:, :, z])
    tmp = tempfile.mkdtemp()
    with contextlib.redirect_stdout(io.StringIO()):
        trk = ref_tracker.Tracker(volume_num=2, siz_xyz=lab.shape, z_xy_ratio=3.0, z_scaling=4, noise_level=100, min_size=20, beta_tk=300,
                                  lambda_tk=0.1, maxiter_tk=20, folder_path=tmp, image_name="img_t%04i_z%04i.tif", unet_model_file="unet.h5",
                                  ffn_model_file="ffn.h5", ensemble=False)
        for f in sorted(d.iterdir()):
            (Path(trk.paths.manual_segmentation_vol1) / f.name).write_bytes(f.read_bytes())
        trk.load_manual_seg()
        relabelled = np.asarray(trk.segmentation_manual_relabels).copy()
        trk.interpolate_seg()
    np.savez_compressed(HERE / "interpolate_seg.npz", para=np.array([3.0, 4]), loaded_relabelled=relabelled.astype(np.int16),
                        seg_interp=np.asarray(trk.seg_cells_interpolated_corrected).astype(np.int16),
                        z_range_interp=np.array(list(trk.Z_RANGE_INTERP)), manual_relabels=np.asarray(trk.segmentation_manual_relabels).astype(np.int16),
                        r_tracked_t0=np.asarray(trk.r_coordinates_tracked_t0, dtype=np.float64), cell_num_t0=np.array(trk.cell_num_t0))
    assert not list(Path("/root/reference").rglob("__pycache__"))
    print("cells", trk.cell_num_t0, "interp shape", trk.seg_cells_interpolated_corrected.shape, "labels loaded", np.unique(relabelled).tolist())


if __name__ == "__main__":
    main()
