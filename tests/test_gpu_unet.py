"""GPU parity: HIP U-Net path (through the C ABI) vs the numpy oracle and the golden vectors."""
import hashlib
import importlib
import json

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import unet_ref as ur

arch_mod = importlib.import_module("3deecelltracker_amd.arch")
synth = importlib.import_module("3deecelltracker_amd.synth")
unet3d = importlib.import_module("3deecelltracker_amd.unet3d")


class FakeUNet:
    def __init__(self, shape):
        self.input_shape = (None, *shape, 1)
        self.output_shape = (None, *shape, 1)
        i, j, k = np.meshgrid(*(np.arange(s) for s in shape), indexing="ij")
        self.ramp = (((i * 31 + j * 17 + k * 7) % 101) / 101.0).astype(np.float32)

    def predict(self, x, **_):
        return (x * np.float32(0.5) + self.ramp[None, :, :, :, None]).astype(np.float32)


def test_tiler_matches_reference_golden(golden_dir):
    """device reflect-gather + centre scatter, driven through unet3_prediction with the same fake
    models the reference was run with -> bit-exact (sha256) against the reference's output."""
    meta = json.loads((golden_dir / "tiler.json").read_text())
    for c in meta:
        if int(np.prod(c["vol"])) > 300 * 300 * 32:
            continue     # 75 host round trips of the fake model: covered by the property test below
        img = np.random.default_rng(c["seed"]).normal(0, 1, (1, *c["vol"], 1)).astype(np.float32)
        res = unet3d.unet3_prediction(img, FakeUNet(tuple(c["net"])), shrink=tuple(c["shrink"]))
        assert res.dtype == np.float32 and res.shape == img.shape
        assert hashlib.sha256(np.ascontiguousarray(res).tobytes()).hexdigest() == c["sha256"], c


def test_tiler_identity_full_size():
    """512x512x32 / 75 patches: gather then scatter of the untouched patches is the identity."""
    import torch
    lib = importlib.import_module("3deecelltracker_amd._lib")
    L = lib.lib()
    vol = torch.randn(512, 512, 32, device="cuda")
    net, shrink = (160, 160, 16), (24, 24, 2)
    centre, grid = unet3d.tile_plan(vol.shape, net, shrink)
    assert grid == (5, 5, 3) and centre == (112, 112, 12)
    patches = torch.empty((75, *net), device="cuda")
    out = torch.zeros_like(vol)
    st = torch.cuda.current_stream().cuda_stream
    lib.check(L.ct_tile_gather_reflect(vol.data_ptr(), lib.ivec(vol.shape), lib.ivec(net), lib.ivec(shrink), 0, 75, patches.data_ptr(), st))
    lib.check(L.ct_tile_scatter_center(patches.data_ptr(), lib.ivec(vol.shape), lib.ivec(net), lib.ivec(shrink), 0, 75, out.data_ptr(), st))
    torch.cuda.synchronize()
    assert torch.equal(out, vol)
    # reflect padding against numpy on one border patch
    ref = np.pad(vol.cpu().numpy(), ((24, 24 + 48), (24, 24 + 48), (2, 2 + 4)), "reflect")
    assert np.array_equal(patches[74].cpu().numpy(), ref[4 * 112:4 * 112 + 160, 4 * 112:4 * 112 + 160, 24:40])


@pytest.mark.parametrize("name", ["unet3_a", "unet3_c", "unet3_b"])
def test_unet_layers_against_oracle(name):
    """every conv block of one patch against the fp32 oracle, then the probability map (<= 1e-4)."""
    import torch
    arch = arch_mod.ARCHS[name]
    w = synth.make_unet_weights(name, seed=1)
    model = getattr(unet3d, name)().set_weights_dict(w)
    patch = np.random.default_rng(2).normal(size=arch.input_shape).astype(np.float32)
    collect = []
    want = ur.unet_forward(patch, w, arch, dtype=np.float32, collect=collect)
    got, dump = model.predict_device(torch.from_numpy(patch[None]).cuda(), layer_dump=True)
    torch.cuda.synchronize()
    dump = dump.cpu().numpy()
    off = 0
    for i, ref in enumerate(collect):
        n = ref.size
        mine = dump[off:off + n].reshape(ref.shape); off += n
        scale = max(1.0, float(np.abs(ref).max()))
        err = float(np.abs(mine - ref).max())
        assert err <= 2e-5 * scale, f"{name} conv block {i}: max abs err {err} (scale {scale})"
    assert off == dump.size
    err = float(np.abs(got[0].cpu().numpy() - want).max())
    assert err <= 1e-4, f"{name} probability map: max abs err {err}"


def test_unet3_prediction_small_volume():
    """BASELINE config 0: 64x64x16 stack (2 patches; reflect pad 72 > extent 64) end to end."""
    arch = arch_mod.UNET3_A
    w = synth.make_unet_weights("unet3_a", seed=0)
    model = unet3d.unet3_a().set_weights_dict(w)
    img = np.random.default_rng(4).normal(size=(1, 64, 64, 16, 1)).astype(np.float32)
    got = unet3d.unet3_prediction(img, model)
    want = ur.unet3_prediction_ref(img, lambda p: ur.unet_forward(p, w, arch), arch.input_shape)
    assert got.shape == img.shape and got.dtype == np.float32
    assert float(np.abs(got - want).max()) <= 1e-4


def test_keras_predict_surface():
    w = synth.make_unet_weights("unet3_c", seed=5)
    model = unet3d.unet3_c().set_weights_dict(w)
    assert model.input_shape == (None, 64, 64, 64, 1) and model.output_shape == (None, 64, 64, 64, 1)
    x = np.random.default_rng(0).normal(size=(2, 64, 64, 64, 1)).astype(np.float32)
    y = model.predict(x)
    assert y.shape == x.shape and y.dtype == np.float32
    y0 = model.predict(x[:1])
    assert np.array_equal(y0, y[:1])          # batching does not change results
    with pytest.raises(ValueError):
        model.predict(x[:, :32])
    with pytest.raises(NotImplementedError):
        model.compile()
    with pytest.raises(ValueError):
        unet3d.unet3_a().predict(np.zeros((1, 160, 160, 16, 1), np.float32))   # no weights loaded
