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


def _full_volume_against_oracle(model, w, arch, vol_xyz, shrink=(24, 24, 2), fp64_patches=(0,)):
    """EVERY patch of a volume: the device's stitched map against the torch-CPU evaluation of the oracle network on all patches
    (fp32 oneDNN convolutions, ~0.1-0.2 s per patch; the numpy oracle needs 3.7 s), after the torch evaluation itself has been held to
    the numpy oracle and to its own fp64 run on `fp64_patches`.  Returns the per-patch maximum error over the kept crops."""
    import torch
    got = model.predict_volume_device(torch.from_numpy(np.ascontiguousarray(vol_xyz)).cuda(), shrink=shrink).cpu().numpy()
    want, pred, plan = ur.unet3_prediction_torch(vol_xyz, w, arch, shrink=shrink, dtype=np.float32)
    patches = ur.gather_patches(vol_xyz, plan)
    for p in fp64_patches:
        ref64 = ur.unet_forward_torch(patches[p], w, arch, dtype=np.float64)
        assert float(np.abs(pred[p] - ref64).max()) <= 2e-6, f"torch fp32 oracle vs torch fp64, patch {p}"
        ref_np = ur.unet_forward(patches[p], w, arch)
        assert float(np.abs(ref_np - ref64).max()) <= 2e-6, f"numpy oracle vs torch fp64, patch {p}"
    gx, gy, gz = plan["grid"]; cx, cy, cz = plan["centre"]
    errs = np.zeros(gx * gy * gz)
    for p in range(len(errs)):
        i, j, k = p // (gy * gz), (p // gz) % gy, p % gz
        sl = (slice(i * cx, (i + 1) * cx), slice(j * cy, (j + 1) * cy), slice(k * cz, (k + 1) * cz))
        d = np.abs(got[sl] - want[sl])
        errs[p] = float(d.max()) if d.size else 0.0
    return errs, plan


def test_config1_256x256x24_all_patches_against_oracle():
    """BASELINE config 1 size (256x256x24 -> 18 patches): all 18 patches against the oracle network."""
    arch = arch_mod.UNET3_A
    w = synth.make_unet_weights("unet3_a", seed=2)
    model = unet3d.unet3_a().set_weights_dict(w)
    vol = np.random.default_rng(7).normal(size=(256, 256, 24)).astype(np.float32)
    errs, plan = _full_volume_against_oracle(model, w, arch, vol, fp64_patches=(0, 9))
    assert plan["grid"] == (3, 3, 2) and len(errs) == 18
    assert errs.max() <= 1e-4, f"worst patch {int(errs.argmax())}: {errs.max()}"


def test_config2_512x512x32_size_independent_properties():
    """BASELINE metric size (75 patches): results do not depend on how the patch range is split (what the multi-GPU
    patch sharding relies on), on the batch size, nor on the run (determinism); probabilities are in (0, 1)."""
    import torch
    model = unet3d.unet3_a().set_weights_dict(synth.make_unet_weights("unet3_a", seed=0))
    vol = torch.from_numpy(synth.normalize_stack(synth.make_stack((512, 512, 32), 600, seed=0)[0])[0, :, :, :, 0].copy()).cuda()
    whole = model.predict_volume_device(vol)
    again = model.predict_volume_device(vol)
    assert torch.equal(whole, again)
    parts = torch.zeros_like(vol)
    for b, e in ((0, 10), (10, 19), (19, 28), (28, 37), (37, 46), (46, 55), (55, 65), (65, 75)):     # 8-rank split
        model.predict_volume_device(vol, p_begin=b, n=e - b, out=parts)
    assert torch.equal(whole, parts)
    small_batches = model.predict_volume_device(vol, max_batch=7)
    assert torch.equal(whole, small_batches)
    assert float(whole.min()) > 0.0 and float(whole.max()) < 1.0 and bool(torch.isfinite(whole).all())


def test_worm4_shape_88_patches_against_oracle():
    """168x401x128 (worm4, BASELINE config 4's stack): 88 patches, 11 patch layers in z, ragged far faces in x and y: all 88 patches
    against the oracle network."""
    arch = arch_mod.UNET3_A
    w = synth.make_unet_weights("unet3_a", seed=0)
    model = unet3d.unet3_a().set_weights_dict(w)
    vol = np.random.default_rng(3).normal(size=(168, 401, 128)).astype(np.float32)
    errs, plan = _full_volume_against_oracle(model, w, arch, vol, fp64_patches=(87,))
    assert len(errs) == 88 and plan["grid"][2] == 11
    assert errs.max() <= 1e-4, f"worst patch {int(errs.argmax())}: {errs.max()}"


def test_tiler_512_golden_with_the_fake_model_on_device(golden_dir):
    """The 512x512x32 / 75-patch golden case (the reference's own unet3_prediction output with the fake model, sha256):
    reflect gather of all 75 patches on the device, the fake model's arithmetic (x * 0.5 + ramp, two fp32 roundings like
    numpy's) applied on the device, centre-crop stitch -> bit-identical to the reference's output."""
    import torch
    lib = importlib.import_module("3deecelltracker_amd._lib"); L = lib.lib()
    meta = [c for c in json.loads((golden_dir / "tiler.json").read_text()) if tuple(c["vol"]) == (512, 512, 32)]
    assert len(meta) == 1
    c = meta[0]
    net, shrink = tuple(c["net"]), tuple(c["shrink"])
    img = np.random.default_rng(c["seed"]).normal(0, 1, (1, *c["vol"], 1)).astype(np.float32)
    vol = torch.from_numpy(np.ascontiguousarray(img[0, :, :, :, 0])).cuda()
    centre, grid = unet3d.tile_plan(vol.shape, net, shrink)
    n = grid[0] * grid[1] * grid[2]
    assert n == 75
    patches = torch.empty((n, *net), device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    lib.check(L.ct_tile_gather_reflect(vol.data_ptr(), lib.ivec(vol.shape), lib.ivec(net), lib.ivec(shrink), 0, n, patches.data_ptr(), st))
    ramp = torch.from_numpy(FakeUNet(net).ramp).cuda()
    pred = (patches * 0.5 + ramp[None]).contiguous()
    out = torch.zeros_like(vol)
    lib.check(L.ct_tile_scatter_center(pred.data_ptr(), lib.ivec(vol.shape), lib.ivec(net), lib.ivec(shrink), 0, n, out.data_ptr(), st))
    res = out.cpu().numpy()[None, :, :, :, None]
    assert hashlib.sha256(np.ascontiguousarray(res).tobytes()).hexdigest() == c["sha256"]
    # the multi-GPU exchange unit: centre crops packed into slabs and unpacked elsewhere reproduce the same volume
    per = centre[0] * centre[1] * centre[2]
    slab = torch.empty((n, per), device="cuda"); back = torch.zeros_like(vol)
    lib.check(L.ct_tile_pack_crops(out.data_ptr(), lib.ivec(vol.shape), lib.ivec(net), lib.ivec(shrink), 0, n, slab.data_ptr(), st))
    for b, e in ((0, 10), (10, 19), (19, 28), (28, 37), (37, 46), (46, 55), (55, 65), (65, 75)):       # 8-rank split
        lib.check(L.ct_tile_unpack_crops(slab[b:e].contiguous().data_ptr(), lib.ivec(vol.shape), lib.ivec(net), lib.ivec(shrink), b, e - b,
                                         back.data_ptr(), st))
    assert torch.equal(back, out)


def test_config2_512x512x32_all_75_patches_against_oracle():
    """BASELINE metric size (512x512x32 -> 75 patches, the benchmark's synthetic stack after LCN): ALL 75 patches of the device volume --
    the 48 full-crop ones and the 27 on the far faces that take the second dependency walk of the crop-aware decoder -- against the
    oracle network."""
    arch = arch_mod.UNET3_A
    w = synth.make_unet_weights("unet3_a", seed=0)
    model = unet3d.unet3_a().set_weights_dict(w)
    stack, _ = synth.make_stack((512, 512, 32), 600, seed=0)
    vol = np.ascontiguousarray(synth.normalize_stack(stack)[0, :, :, :, 0])
    errs, plan = _full_volume_against_oracle(model, w, arch, vol, fp64_patches=(0, 74))
    assert plan["grid"] == (5, 5, 3) and len(errs) == 75
    assert errs.max() <= 1e-4, f"worst patch {int(errs.argmax())}: {errs.max()}"
    # the Keras-facing entry point returns the same map
    got = unet3d.unet3_prediction(vol[None, :, :, :, None], model)[0, :, :, :, 0]
    import torch
    assert np.array_equal(got, model.predict_volume_device(torch.from_numpy(vol).cuda()).cpu().numpy())


def test_split_fp16_scaling_over_the_dynamic_range():
    """The default conv family splits fp32 operands into fp16 pairs after an exact per-patch power-of-two scaling by the input
    tensor's maximum.  One batch with patches of very different magnitude (x 1e-4, x 1, x 1e4), an all-zero patch and a patch with
    a single 1e5 outlier: every conv block of every patch stays within 2e-5 of ITS OWN scale vs the fp64-accumulating oracle, the
    patches do not influence each other (bit-identical to running them alone), nothing overflows."""
    import torch
    arch = arch_mod.UNET3_A
    w = synth.make_unet_weights("unet3_a", seed=5)
    model = unet3d.unet3_a().set_weights_dict(w)
    rng = np.random.default_rng(6)
    base = rng.normal(size=arch.input_shape).astype(np.float32)
    outlier = (0.01 * rng.normal(size=arch.input_shape)).astype(np.float32); outlier[80, 80, 8] = 1e5
    batch = np.stack([base * 1e-4, base, base * 1e4, np.zeros_like(base), outlier]).astype(np.float32)
    got = model.predict_device(torch.from_numpy(batch).cuda()).cpu().numpy()
    assert np.isfinite(got).all()
    for p in range(len(batch)):
        collect = []
        want = ur.unet_forward(batch[p], w, arch, dtype=np.float32, collect=collect)
        alone, dump = model.predict_device(torch.from_numpy(batch[p:p + 1]).cuda(), layer_dump=True)
        assert np.array_equal(alone[0].cpu().numpy(), got[p]), f"patch {p}: result depends on its batch neighbours"
        dump = dump.cpu().numpy(); off = 0
        for i, ref in enumerate(collect):
            mine = dump[off:off + ref.size].reshape(ref.shape); off += ref.size
            scale = max(float(np.abs(ref).max()), 1e-30)
            err = float(np.abs(mine - ref).max())
            assert err <= 2e-5 * scale, f"patch {p} conv block {i}: max abs err {err} at scale {scale}"
        # the head is a sigmoid of a logit whose error scales with the last block's magnitude (slope <= 1/4): 1e-4 at ordinary scale
        tol = max(1e-4, 0.25 * 2e-5 * float(np.abs(collect[-1]).max()) * float(np.abs(w["head"]["kernel"]).sum()))
        assert float(np.abs(got[p] - want).max()) <= tol, f"patch {p}: probability map (tolerance {tol})"


@pytest.mark.parametrize("amp_log2", (12, 20))
def test_split_fp16_with_batchnorm_statistics_that_amplify_one_channel(amp_log2):
    """The split's precision is relative to the per-patch TENSOR maximum.  A trained net can hold a channel whose BatchNorm scale dwarfs
    the others': here one channel of two tensors (the first conv of level 1; the first conv of the last decoder level) is amplified by
    2^12 / 2^20 through gamma and beta, and the single consumer's weights for it are divided by the same power of two -- the network
    function is unchanged in exact arithmetic, but every OTHER channel of those tensors now sits 2^12 / 2^20 below the tensor maximum.
    Values down to 2^-17 of the maximum keep full precision, below that the absolute error is 2^-39 of the maximum: at
    2^20 that is 2^-19 relative to O(1) activations.  The probability map must stay within 1e-4 of the fp64 oracle of the SAME weights
    and of the unamplified network, on the patch path and on the (fused, crop-aware) volume path."""
    import copy
    import torch
    arch = arch_mod.UNET3_A
    w0 = synth.make_unet_weights("unet3_a", seed=9)
    w = copy.deepcopy(w0)
    amp = np.float32(2.0 ** amp_log2)
    for layer, ch in ((2, 5), (12, 3)):                      # tensors with exactly one consumer: conv `layer + 1`
        w["convs"][layer]["gamma"][ch] *= amp; w["convs"][layer]["beta"][ch] *= amp
        w["convs"][layer]["mean"] = w["convs"][layer]["mean"].copy()
        w["convs"][layer + 1]["kernel"][:, :, :, ch, :] /= amp
    rng = np.random.default_rng(10)
    patch = rng.normal(size=arch.input_shape).astype(np.float32)
    want = ur.unet_forward_torch(patch, w, arch, dtype=np.float64)
    base = ur.unet_forward_torch(patch, w0, arch, dtype=np.float64)
    assert float(np.abs(want - base).max()) <= 1e-9           # the same function
    m_amp = unet3d.unet3_a().set_weights_dict(w)
    m_ref = unet3d.unet3_a().set_weights_dict(w0)
    got = m_amp.predict_device(torch.from_numpy(patch[None]).cuda())[0].cpu().numpy()
    ref = m_ref.predict_device(torch.from_numpy(patch[None]).cuda())[0].cpu().numpy()
    assert np.isfinite(got).all()
    assert float(np.abs(got - want).max()) <= 1e-4, float(np.abs(got - want).max())
    assert float(np.abs(got - ref).max()) <= 1e-4
    vol = torch.from_numpy(rng.normal(size=(200, 180, 20)).astype(np.float32)).cuda()
    assert float((m_amp.predict_volume_device(vol) - m_ref.predict_volume_device(vol)).abs().max()) <= 1e-4


@pytest.mark.parametrize("case", ("x1e-4", "x1", "x1e4", "amplified_first_conv"))
def test_fused_first_pair_scales_by_a_bound_of_the_first_convs_outputs(case):
    """Volume path (conv_l0l1_fused_kernel, round 6): the second conv's input scale is taken from a BOUND of the first conv's outputs
    (bound_a * max|input tile| + bound_b, host constants) instead of their measured maximum.  The bound exceeds the true maximum by the slack of
    the triangle inequality; the split keeps full precision 2^17 below its scale, so a few bits of slack must not show: one-patch volumes whose
    input is scaled by 1e-4 / 1 / 1e4, and a net whose first conv has one channel amplified by 2^12 through its BatchNorm (the consumer's
    weights divided by the same power of two: the same function, a bound 2^12 above every other channel), against the fp64 oracle."""
    import copy
    import torch
    arch = arch_mod.UNET3_A
    w = synth.make_unet_weights("unet3_a", seed=21)
    scale = {"x1e-4": 1e-4, "x1": 1.0, "x1e4": 1e4}.get(case, 1.0)
    if case == "amplified_first_conv":
        w = copy.deepcopy(w)
        amp = np.float32(2.0 ** 12)
        w["convs"][0]["gamma"][3] *= amp; w["convs"][0]["beta"][3] *= amp
        w["convs"][1]["kernel"][:, :, :, 3, :] /= amp
    rng = np.random.default_rng(22)
    vol = (scale * rng.normal(size=(112, 112, 12))).astype(np.float32)          # one 160 x 160 x 16 patch after padding by (24, 24, 2)
    model = unet3d.unet3_a().set_weights_dict(w)
    got = model.predict_volume_device(torch.from_numpy(vol).cuda()).cpu().numpy()
    last = []

    def oracle_patch(p):
        collect = []
        out = ur.unet_forward(p, w, arch, dtype=np.float32, collect=collect)          # fp64-accumulating oracle of the fp32 network
        last.append(float(np.abs(collect[-1]).max()))
        return out
    want = ur.unet3_prediction_ref(vol[None, :, :, :, None], oracle_patch, arch.input_shape)[0, :, :, :, 0]
    assert np.isfinite(got).all()
    # the head is a sigmoid of a logit whose error scales with the last block's magnitude (slope <= 1/4): 1e-4 at ordinary scale -- the tolerance of
    # test_split_fp16_scaling_over_the_dynamic_range, which holds the patch path to the same inputs
    tol = max(1e-4, 0.25 * 2e-5 * max(last) * float(np.abs(w["head"]["kernel"]).sum()))
    err = float(np.abs(got - want).max())
    assert err <= tol, (case, err, tol)
    # and the fused path is no worse than the two-kernel path it replaces: the patch entry point (conv_first_f16_kernel + the plain second conv) on the same patch
    plan_patch = ur.gather_patches(vol, ur.tile_plan(vol.shape, arch.input_shape, arch.input_shape, (24, 24, 2)))[0].astype(np.float32)
    two = model.predict_device(torch.from_numpy(plan_patch[None]).cuda())[0].cpu().numpy()
    want_patch = ur.unet_forward(plan_patch, w, arch, dtype=np.float32)
    err_two = float(np.abs(two - want_patch).max())
    assert err <= max(tol, 4.0 * err_two), (case, "fused", err, "two kernels", err_two)


def test_volume_path_computes_only_what_the_centre_crops_need():
    """ct_unet_predict_volume evaluates decoder tiles only where a kept (centre-crop) voxel depends on them; the rest of its
    workspace is never read by a kept voxel.  (1) The stitched map equals -- within the arithmetic's own noise -- the one
    assembled from fully evaluated patches; (2) it does not change when the workspace is filled with NaN / huge values
    beforehand (uncomputed regions must not leak into kept voxels, nor into the per-patch scale exponents); (3) the computed
    fractions are what the dependency walk predicts for unet3_a with shrink (24, 24, 2)."""
    import ctypes as C
    import torch
    arch = arch_mod.UNET3_A
    model = unet3d.unet3_a().set_weights_dict(synth.make_unet_weights("unet3_a", seed=7))
    vol = torch.randn(200, 230, 20, device="cuda")
    ref_out = model.predict_volume_device(vol)
    # (1) against patch-by-patch full evaluation + the reference's stitching
    plan = ur.tile_plan(tuple(vol.shape), arch.input_shape, arch.input_shape, (24, 24, 2))
    patches = ur.gather_patches(vol.cpu().numpy(), plan)
    full = model.predict_device(torch.from_numpy(np.ascontiguousarray(patches)).cuda()).cpu().numpy()
    want = ur.scatter_centres(full, plan, tuple(vol.shape))
    assert float(np.abs(ref_out.cpu().numpy() - want).max()) <= 5e-6
    # (2) poison the cached workspace, run again: bit-identical
    _lib = importlib.import_module("3deecelltracker_amd._lib")
    ws = model._workspace(_lib.lib().ct_unet_workspace_bytes(model._handle, 128))
    for poison in (float("nan"), 3.0e38, -1.0e30):
        ws.view(torch.float32).fill_(poison)
        assert torch.equal(model.predict_volume_device(vol), ref_out)
    # (3) the dependency walk
    L = _lib.lib()
    nl = L.ct_unet_num_conv_layers(model._handle)
    frac = []
    for i in range(nl):
        reg = (C.c_int * 4)(); d = (C.c_int * 3)()
        L.ct_unet_layer_region(model._handle, i, reg); L.ct_unet_layer_info(model._handle, i, None, None, d, None)
        frac.append((reg[1] - reg[0]) * (reg[3] - reg[2]) / float(d[0] * d[1]))
    assert all(f == 1.0 for f in frac[:8]) and frac[13] == (112 * 112) / (160 * 160) and frac[12] == (116 * 120) / (160 * 160)
    assert frac[8] <= 1.0 and frac[9] < 1.0 and frac[10] < 0.7 and frac[11] < 0.7, frac
    model.predict_device(torch.from_numpy(np.ascontiguousarray(patches[:1])).cuda())        # the patch entry point computes everything
    L.ct_unet_layer_region(model._handle, 13, reg)
    assert tuple(reg) == (0, 160, 0, 160)


@pytest.mark.parametrize("name,shape,shrink", [("unet3_a", (112, 112, 12), (24, 24, 2)), ("unet3_a", (224, 250, 12), (24, 24, 2)),
                                               ("unet3_a", (300, 120, 30), (16, 8, 2)), ("unet3_c", (100, 90, 70), (12, 12, 12)),
                                               ("unet3_c", (64, 64, 64), (8, 10, 6))])
def test_volume_path_ignores_uncomputed_workspace(name, shape, shrink):
    """The crop-aware decoder leaves parts of its tensors unwritten; nothing a kept voxel depends on may read them -- not even
    through a zero weight (0 x NaN).  Workspace filled with NaN / Inf / huge values before the run: bit-identical output, for volumes
    with and without partial crops on their far faces, other shrink values, and the net with (2, 2, 2) pools and several z blocks."""
    import torch
    model = getattr(unet3d, name)().set_weights_dict(synth.make_unet_weights(name, seed=8))
    vol = torch.randn(*shape, device="cuda")
    ref_out = model.predict_volume_device(vol, shrink=shrink).clone()
    assert bool(torch.isfinite(ref_out).all())
    _lib = importlib.import_module("3deecelltracker_amd._lib")
    ws = model._workspace(_lib.lib().ct_unet_workspace_bytes(model._handle, 128))
    for poison in (float("nan"), float("inf"), -3.0e38):
        ws.view(torch.float32).fill_(poison)
        assert torch.equal(model.predict_volume_device(vol, shrink=shrink), ref_out), f"{name} {shape} {shrink}: poison {poison} leaked"


@pytest.mark.gpu
def test_volume_path_writes_nothing_outside_its_buffers():
    """Workspace, input and output between 64-MB guard bands (scripts/probe/guard_probe.py): no byte of a band may change."""
    import subprocess
    import sys
    from pathlib import Path
    repo = Path(__file__).resolve().parent.parent
    r = subprocess.run([sys.executable, str(repo / "scripts" / "probe" / "guard_probe.py"), "unet3_a"], capture_output=True, text=True,
                       timeout=600, cwd=repo)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "guard bytes overwritten: 0" in r.stdout, r.stdout[-1000:]
