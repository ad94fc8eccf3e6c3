"""Pins oracle/unet_ref.py: tiler against the reference's own unet3_prediction (golden), layer
arithmetic against an independent torch-CPU float64 evaluation.  CPU only."""
import hashlib
import importlib
import json

import numpy as np
import pytest

from oracle import unet_ref as ur

arch_mod = importlib.import_module("3deecelltracker_amd.arch")
synth = importlib.import_module("3deecelltracker_amd.synth")


def _fake_predict(net):
    i, j, k = np.meshgrid(*(np.arange(s) for s in net), indexing="ij")
    ramp = (((i * 31 + j * 17 + k * 7) % 101) / 101.0).astype(np.float32)
    return lambda p: p * np.float32(0.5) + ramp


def test_tiler_against_reference(golden_dir):
    meta = json.loads((golden_dir / "tiler.json").read_text())
    g = np.load(golden_dir / "tiler.npz")
    for idx, c in enumerate(meta):
        rng = np.random.default_rng(c["seed"])
        img = rng.normal(0, 1, (1, *c["vol"], 1)).astype(np.float32)
        res = ur.unet3_prediction_ref(img, _fake_predict(c["net"]), tuple(c["net"]), shrink=tuple(c["shrink"]))
        assert res.dtype == np.float32 and res.shape == img.shape
        assert hashlib.sha256(np.ascontiguousarray(res).tobytes()).hexdigest() == c["sha256"], c
        if f"tiler_out_{idx}" in g:
            assert np.array_equal(res[0, :, :, :, 0], g[f"tiler_out_{idx}"])


def test_identity_model_is_exact():
    img = np.random.default_rng(0).normal(size=(1, 64, 64, 16, 1)).astype(np.float32)  # pad 72 > extent 64
    res = ur.unet3_prediction_ref(img, lambda p: p, (160, 160, 16))
    assert np.array_equal(res, img)


def test_reflect_index_matches_numpy_pad():
    for n, before, after in ((5, 3, 4), (5, 9, 11), (2, 5, 5), (1, 3, 2), (64, 24, 72)):
        a = np.arange(n)
        want = np.pad(a, (before, after), "reflect")
        got = ur.reflect_index(np.arange(-before, n + after), n)
        assert np.array_equal(want, got), (n, before, after)


def test_patch_counts():
    a = arch_mod.UNET3_A
    for vol, want in (((64, 64, 16), 2), ((256, 256, 24), 18), ((512, 512, 32), 75), ((168, 401, 128), 88)):
        plan = ur.tile_plan(vol, a.input_shape, a.input_shape, (24, 24, 2))
        assert int(np.prod(plan["grid"])) == want


@pytest.mark.parametrize("name,shape", [("unet3_a", (16, 24, 8)), ("unet3_c", (16, 16, 24)), ("unet3_b", (12, 8, 4))])
def test_unet_forward_against_torch_fp64(name, shape):
    pytest.importorskip("torch")
    arch = arch_mod.ARCHS[name]
    w = synth.make_unet_weights(name, seed=3)
    patch = np.random.default_rng(5).normal(size=shape).astype(np.float32)
    ref = ur.unet_forward_torch(patch, w, arch, dtype=np.float64)
    got64 = ur.unet_forward(patch, w, arch, dtype=np.float64)
    np.testing.assert_allclose(got64, ref, rtol=0, atol=1e-12)
    got32 = ur.unet_forward(patch, w, arch, dtype=np.float32)
    np.testing.assert_allclose(got32, ref, rtol=0, atol=2e-5)
    np.testing.assert_allclose(ur.unet_forward_torch(patch, w, arch, dtype=np.float32), ref, rtol=0, atol=2e-5)   # the CPU-baseline path


def test_arch_work_figures():
    a = arch_mod.UNET3_A
    assert abs(a.flops_per_patch() / 1e9 - 35.57) < 0.01
    assert abs(a.algorithmic_bytes_per_patch() / 1e6 - 285.1) < 0.1
    assert abs(arch_mod.UNET3_B.flops_per_patch() / 1e9 - 196.0) < 0.1
    assert abs(arch_mod.UNET3_C.flops_per_patch() / 1e9 - 12.1) < 0.05
