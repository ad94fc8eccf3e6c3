"""The ctypes stubs printed in INTEGRATION.md section B are executed as written (only the library path is substituted): the documentation
must not drift from include/ctamd.h."""
import importlib
import re
import types
from pathlib import Path

import numpy as np
import pytest

REPO = Path(__file__).resolve().parent.parent
_lib = importlib.import_module("3deecelltracker_amd._lib")


def _blocks():
    text = (REPO / "INTEGRATION.md").read_text()
    sec = text[text.index("## B. Surgical"):]
    return re.findall(r"```python\n(.*?)```", sec, flags=re.S)


def test_stub_code_parses():
    blocks = _blocks()
    assert len(blocks) >= 2 and "ct_unet_predict_volume" in blocks[0] and "ct_watershed_segment" in blocks[1]
    for b in blocks[:2]:
        compile(b, "INTEGRATION.md", "exec")


@pytest.mark.gpu
def test_stubs_run_against_the_library():
    from oracle import unet_ref as ur
    from oracle import watershed_ref as wr
    synth = importlib.import_module("3deecelltracker_amd.synth")
    unet3d = importlib.import_module("3deecelltracker_amd.unet3d")
    ns = {}
    blocks = _blocks()
    exec(blocks[0].replace('"libctamd.so"', repr(str(_lib.LIB_PATH))), ns)
    exec(blocks[1], ns)
    # watershed stub == oracle
    tw = importlib.import_module("test_watershed")
    prob = tw.touching_case()
    me = types.SimpleNamespace(z_xy_ratio=3.0, min_size=40, cell_num=0)
    labels = ns["hip_watershed"](me, prob[None, :, :, :, None], "min_size")
    want, _, ms, cn = wr.segment_centroids(prob, 3.0, "min_size", 40)
    assert np.array_equal(labels, want) and (me.min_size, me.cell_num) == (ms, cn)
    # U-Net stub == the host mirror (same library underneath) on a ragged volume
    w = synth.make_unet_weights("unet3_a", seed=3)
    model = unet3d.unet3_a().set_weights_dict(w)
    vol = np.random.default_rng(0).normal(size=(180, 170, 18)).astype(np.float32)
    got = ns["HipUNet"](model.arch.arch_id, unet3d.flatten_unet_weights(w), model.arch.input_shape).predict_volume(vol, (24, 24, 2))
    ref = unet3d.unet3_prediction(vol[None, :, :, :, None], model)[0, :, :, :, 0]
    assert np.array_equal(got, ref)
