"""Data-format edge of the legacy Tracker: per-layer TIFF files (reference tracker.py:113-142 reads them with tifffile, this package with
PIL).  The fixtures were written by tifffile and the expected arrays are what the REFERENCE's read_image_ts returned for them
(tests/golden/make_tiff_fixtures.py, run under the image's second interpreter)."""
import importlib
import os
import subprocess

import numpy as np
import pytest

trk = importlib.import_module("3deecelltracker_amd.tracker")


def test_read_image_ts_equals_the_reference_on_tifffile_written_slices(golden_dir):
    want = np.load(golden_dir / "tiff_expected.npz")
    d = str(golden_dir / "tiff") + "/"
    raw = trk.read_image_ts(2, d, "raw_t%04i_z%04i.tif", (1, 4))
    lab = trk.read_image_ts(2, d, "lab_t%04i_z%04i.tif", (1, 4))
    assert raw.dtype == want["raw"].dtype and np.array_equal(raw, want["raw"])          # (row, column, layer), uint16 incl. 0 and 65535
    assert lab.dtype == want["lab"].dtype and np.array_equal(lab, want["lab"])


def test_written_label_slices_are_read_back_by_tifffile(tmp_path):
    """save_automatic_segmentation (reference :145-165) writes with PIL here; the reference's tools read with tifffile."""
    py = "/opt/conda/bin/python3.9"
    if not os.path.exists(py):
        pytest.skip("no interpreter with tifffile on this machine")
    rng = np.random.default_rng(0)
    for use8, hi in ((True, 200), (False, 3000)):
        labels = rng.integers(0, hi, size=(8, 6, 3)).astype(np.int32)
        out = tmp_path / f"u{int(use8)}"
        trk.save_automatic_segmentation(labels, str(out), use8)
        np.save(out / "want.npy", labels)
        code = ("import numpy as np, tifffile, sys; d = sys.argv[1]; want = np.load(d + '/want.npy'); "
                "got = np.stack([tifffile.imread(d + '/auto_vol1/auto_vol1_z%04i.tif' % z) for z in range(1, 4)], -1); "
                "assert got.dtype == (np.uint8 if sys.argv[2] == '1' else np.uint16), got.dtype; assert np.array_equal(got, want); print('ok')")
        r = subprocess.run([py, "-W", "ignore", "-c", code, str(out), str(int(use8))], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-500:]
