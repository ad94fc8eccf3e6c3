#!/usr/bin/env python3
"""TIFF slices as an acquisition system writes them (tifffile, uint16 and uint8, one file per z layer) and what THE REFERENCE'S OWN
`read_image_ts` (CellTracker/tracker.py:113-142, real tifffile) returns for them.  Runs under the image's second interpreter, which has
tifffile (the main one does not; the package reads TIFF with PIL):

    PYTHONDONTWRITEBYTECODE=1 /opt/conda/bin/python3.9 -W ignore tests/golden/make_tiff_fixtures.py

Writes tests/golden/tiff/*.tif (data files, < 2 KB together) and tests/golden/tiff_expected.npz."""
import importlib
import sys
from pathlib import Path
from unittest.mock import MagicMock

sys.dont_write_bytecode = True
HERE = Path(__file__).resolve().parent
sys.path.insert(0, "/root/reference")
import numpy as np  # noqa: E402
import tifffile  # noqa: E402


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
    d = HERE / "tiff"
    d.mkdir(exist_ok=True)
    rng = np.random.default_rng(3)
    raw = rng.integers(0, 65536, size=(3, 9, 7), dtype=np.uint16)          # (layer, row, column): non-square on purpose
    raw[0, 0, 0] = 65535; raw[1, 0, 1] = 0
    lab = rng.integers(0, 200, size=(3, 9, 7), dtype=np.uint8)
    for z in range(3):
        tifffile.imwrite(d / ("raw_t%04i_z%04i.tif" % (2, z + 1)), raw[z])
        tifffile.imwrite(d / ("lab_t%04i_z%04i.tif" % (2, z + 1)), lab[z])
    got_raw = ref_tracker.read_image_ts(2, str(d) + "/", "raw_t%04i_z%04i.tif", (1, 4))
    got_lab = ref_tracker.read_image_ts(2, str(d) + "/", "lab_t%04i_z%04i.tif", (1, 4))
    assert got_raw.shape == (9, 7, 3) and got_raw.dtype == np.uint16
    np.savez_compressed(HERE / "tiff_expected.npz", raw=got_raw, lab=got_lab, tifffile_version=np.array(tifffile.__version__))
    assert not list(Path("/root/reference").rglob("__pycache__"))
    print("written", sorted(p.name for p in d.iterdir()), got_raw.dtype, got_lab.dtype)


if __name__ == "__main__":
    main()
