"""The conv kernel families must agree with each other: f32-input MFMA vs split-bf16 (bf16x6) vs split-fp16 (f16x3, the default),
decoder tap folding on/off, Z8 tiles on/off, 4 x 10 tiles never / where they fit better (default) / on every layer that can take them.

The family is chosen when the model is created (CT_CONV_MATH / CT_CONV_FOLD, read once per process), so every mode runs in
its own child process on the same seeded patches (CT_CONV_MATH / CT_CONV_FOLD / CT_CONV_Z8); the default mode is additionally held to the oracle in test_gpu_unet.py."""
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

REPO = Path(__file__).resolve().parent.parent

CHILD = r"""
import importlib, sys
import numpy as np, torch
sys.path.insert(0, sys.argv[1])
name, out = sys.argv[2], sys.argv[3]
synth = importlib.import_module("3deecelltracker_amd.synth")
unet3d = importlib.import_module("3deecelltracker_amd.unet3d")
arch = importlib.import_module("3deecelltracker_amd.arch").ARCHS[name]
model = getattr(unet3d, name)().set_weights_dict(synth.make_unet_weights(name, seed=3))
patches = np.random.default_rng(4).normal(size=(2,) + tuple(arch.input_shape)).astype(np.float32)
got, dump = model.predict_device(torch.from_numpy(patches[:1]).cuda(), layer_dump=True)
both = model.predict_device(torch.from_numpy(patches).cuda())
torch.cuda.synchronize()
np.savez(out, prob=both.cpu().numpy(), dump=dump.cpu().numpy())
"""


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["unet3_a", "unet3_b", "unet3_c"])
def test_kernel_families_agree(name, tmp_path):
    res = {}
    # (math, fold, z8[, y10]): z8 = the 8 x 8 x 8 tile geometry of levels with Z <= 8 (split-bf16 kernels only); y10 = CT_CONV_Y10
    for math, fold, z8, *y10 in (("f32", "0", "1"), ("f32", "1", "1"), ("bf16x6", "0", "1"), ("bf16x6", "1", "1"), ("bf16x6", "1", "0"),
                                 ("f16x3", "0", "1"), ("f16x3", "1", "1"), ("f16x3", "1", "0"), ("f16x3", "1", "1", "0"),
                                 ("f16x3", "1", "1", "0xffff"), ("f16x3", "0", "0", "0xffff")):
        out = tmp_path / f"{name}_{math}_{fold}_{z8}_{''.join(y10)}.npz"
        env = dict(os.environ, CT_CONV_MATH=math, CT_CONV_FOLD=fold, CT_CONV_Z8=z8)
        env.pop("CT_CONV_Y10", None)
        if y10:
            env["CT_CONV_Y10"] = y10[0]
        r = subprocess.run([sys.executable, "-c", CHILD, str(REPO), name, str(out)], env=env, capture_output=True, text=True,
                           timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        res[(math, fold, z8) + tuple(y10)] = np.load(out)
    ref = res[("f32", "0", "1")]                      # the exact-fp32 fmaf-chain kernels without any tap folding
    scale = max(1.0, float(np.abs(ref["dump"]).max()))
    for key, z in res.items():
        assert np.isfinite(z["prob"]).all()
        derr = float(np.abs(z["dump"] - ref["dump"]).max())
        perr = float(np.abs(z["prob"] - ref["prob"]).max())
        assert derr <= 1e-5 * scale, f"{name} {key}: conv blocks differ from the f32 kernels by {derr} (scale {scale})"
        assert perr <= 5e-6, f"{name} {key}: probability maps differ by {perr}"


VOLUME_CHILD = r"""
import importlib, sys
import numpy as np, torch
sys.path.insert(0, sys.argv[1])
out = sys.argv[2]
synth = importlib.import_module("3deecelltracker_amd.synth")
unet3d = importlib.import_module("3deecelltracker_amd.unet3d")
model = unet3d.unet3_a().set_weights_dict(synth.make_unet_weights("unet3_a", seed=3))
vol = torch.from_numpy(np.random.default_rng(5).normal(size=(200, 330, 21)).astype(np.float32)).cuda()
pv = model.predict_volume_device(vol)
torch.cuda.synchronize()
np.save(out, pv.cpu().numpy())
"""


@pytest.mark.gpu
def test_tile_shapes_agree_bitwise_on_a_volume_with_far_face_patches(tmp_path):
    """The 4 x 10 x 16 tiles (CT_CONV_Y10: never / where they cover a level with fewer columns -- the default / every layer that can take them)
    change which workgroup computes a voxel, never how: the volume path (decoder windows, shorter kept crops on the volume's far faces, store
    windows) must give the same bits in all three settings."""
    res = {}
    for y10 in ("0", None, "0xffff"):
        out = tmp_path / f"vol_{y10}.npy"
        env = dict(os.environ)
        env.pop("CT_CONV_Y10", None)
        if y10 is not None:
            env["CT_CONV_Y10"] = y10
        r = subprocess.run([sys.executable, "-c", VOLUME_CHILD, str(REPO), str(out)], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        res[y10] = np.load(out)
    assert np.isfinite(res["0"]).all() and res["0"].std() > 0
    assert np.array_equal(res["0"], res[None]), "default tile selection differs from the 4 x 8 tiles"
    assert np.array_equal(res["0"], res["0xffff"]), "4 x 10 tiles on every eligible layer differ from the 4 x 8 tiles"
