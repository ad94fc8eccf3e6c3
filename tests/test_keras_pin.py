"""Oracle and HIP path against per-block outputs recorded from Keras itself (tests/golden/keras_layers.README).
Skipped while tests/golden/keras_layer_outputs.npz has not been supplied (tensorflow cannot run in the build image)."""
import importlib
from pathlib import Path

import numpy as np
import pytest

from oracle import match_ref as mr
from oracle import preprocess_ref as pr
from oracle import unet_ref as ur

PIN = Path(__file__).resolve().parent / "golden" / "keras_layer_outputs.npz"
pytestmark = pytest.mark.skipif(not PIN.exists(), reason="keras_layer_outputs.npz not supplied: see tests/golden/keras_layers.README")
synth = importlib.import_module("3deecelltracker_amd.synth")
ARCHS = importlib.import_module("3deecelltracker_amd.arch").ARCHS


@pytest.fixture(scope="module")
def k():
    return np.load(PIN)


def _blocks(name, k):
    i = 0
    while f"{name}_block{i}" in k.files:
        yield i, k[f"{name}_block{i}"]; i += 1


@pytest.mark.parametrize("name", ("unet3_a", "unet3_c"))
def test_oracle_unet_against_keras(k, name):
    arch = ARCHS[name]; w = synth.make_unet_weights(name, seed=1)
    patch = np.random.default_rng(2).normal(size=arch.input_shape).astype(np.float32)
    collect = []
    prob = ur.unet_forward(patch, w, arch, dtype=np.float32, collect=collect)
    for i, want in _blocks(name, k):
        got = collect[i][::2, ::2]
        assert got.shape == want.shape
        assert np.abs(got - want).max() <= 2e-5 * max(1.0, float(np.abs(want).max())), f"{name} conv block {i}"
    assert np.abs(prob - k[f"{name}_prob"]).max() <= 1e-5


def test_oracle_ffn_and_lcn_against_keras(k):
    fw = synth.make_ffn_weights(seed=0, gain=6.0, shift=-3.0)
    x = np.random.default_rng(3).normal(size=(4096, 122)).astype(np.float32)
    assert np.abs(mr.ffn_forward(fw, x)[:, 0] - k["ffn_scores"]).max() <= 2e-5
    img = np.random.default_rng(21).integers(0, 3000, size=(64, 64, 16)).astype(np.float64)
    assert np.abs(pr.lcn(img, 100.0, (27, 27, 1), mode="constant") - k["lcn_gpu"]).max() <= 2e-4


@pytest.mark.gpu
@pytest.mark.parametrize("name", ("unet3_a", "unet3_c"))
def test_device_unet_against_keras(k, name):
    import torch
    unet3d = importlib.import_module("3deecelltracker_amd.unet3d")
    arch = ARCHS[name]; w = synth.make_unet_weights(name, seed=1)
    patch = np.random.default_rng(2).normal(size=arch.input_shape).astype(np.float32)
    model = getattr(unet3d, name)().set_weights_dict(w)
    got, dump = model.predict_device(torch.from_numpy(patch[None]).cuda(), layer_dump=True)
    dump = dump.cpu().numpy(); off = 0
    for i, want in _blocks(name, k):
        n = want.shape[0] * 2 * want.shape[1] * 2 * want.shape[2] * want.shape[3]
        full = dump[off:off + n].reshape(want.shape[0] * 2, want.shape[1] * 2, want.shape[2], want.shape[3]); off += n
        assert np.abs(full[::2, ::2] - want).max() <= 1e-4 * max(1.0, float(np.abs(want).max())), f"{name} conv block {i}"
    assert np.abs(got[0].cpu().numpy() - k[f"{name}_prob"]).max() <= 1e-4
