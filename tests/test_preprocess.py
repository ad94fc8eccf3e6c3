"""Local contrast normalisation: oracle vs the reference's lcn_cpu (golden), HIP vs oracle."""
import importlib

import numpy as np
import pytest

from oracle import preprocess_ref as pr

pre = importlib.import_module("3deecelltracker_amd.preprocess")
synth = importlib.import_module("3deecelltracker_amd.synth")


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(golden_dir / "preprocess.npz")


def test_oracle_reflect_variant_against_reference_lcn_cpu(g):
    for i in range(3):
        got = pr.lcn(g[f"lcn_in_{i}"], float(g[f"lcn_nl_{i}"]), tuple(int(v) for v in g[f"lcn_fs_{i}"]), mode="reflect")
        np.testing.assert_allclose(got, g[f"lcn_out_{i}"], rtol=0, atol=1e-12)


def test_oracle_zero_variant_matches_direct_window_sums():
    rng = np.random.default_rng(0)
    x = rng.uniform(0, 100, (9, 11, 3))
    s = pr.box_sum(x, (5, 3, 1), "constant")
    want = np.zeros_like(x)
    for i in range(9):
        for j in range(11):
            want[i, j] = x[max(0, i - 2):i + 3, max(0, j - 1):j + 2].sum(axis=(0, 1))
    np.testing.assert_allclose(s, want, rtol=0, atol=1e-10)
    img = rng.integers(0, 500, (20, 20, 4)).astype(np.uint16)
    out = pr.normalize_image(img, 20.0)
    assert out.shape == img.shape and np.isfinite(out).all()


@pytest.mark.gpu
def test_device_lcn_reflect_against_reference_golden(g):
    for i in range(3):
        got = pre.lcn_cpu(g[f"lcn_in_{i}"], float(g[f"lcn_nl_{i}"]), tuple(int(v) for v in g[f"lcn_fs_{i}"]))
        assert got.dtype == np.float32
        np.testing.assert_allclose(got, g[f"lcn_out_{i}"], rtol=0, atol=2e-4)         # fp32 storage of sums ~1e6..1e8


@pytest.mark.gpu
def test_device_median_is_exact():
    import torch
    rng = np.random.default_rng(1)
    for n in (1, 2, 7, 1000, 1001, 65536 * 3 + 5):
        a = rng.integers(0, 5000, n).astype(np.uint16)
        assert pre.median_device(torch.from_numpy(a).cuda()) == float(np.median(a))
        f = rng.normal(0, 100, n).astype(np.float32)
        assert pre.median_device(torch.from_numpy(f).cuda()) == float(np.median(f.astype(np.float64)))
    z = np.zeros(100, np.uint16); z[:50] = 7
    assert pre.median_device(torch.from_numpy(z).cuda()) == float(np.median(z)) == 3.5
    # the two middle order statistics differ already in the first radix digit (their selections run in the same passes)
    z = np.zeros(1000, np.uint16); z[:500] = 65535
    assert pre.median_device(torch.from_numpy(z).cuda()) == float(np.median(z)) == 32767.5
    z = np.array([255, 256] * 8, np.uint16)
    assert pre.median_device(torch.from_numpy(z).cuda()) == 255.5
    f = np.concatenate([np.full(64, -3.0e20, np.float32), np.full(64, 2.5e-12, np.float32)])
    assert pre.median_device(torch.from_numpy(f).cuda()) == float(np.median(f.astype(np.float64)))
    f = np.array([-1.0, 1.0, -2.0, 2.0], np.float32)
    assert pre.median_device(torch.from_numpy(f).cuda()) == 0.0
    # views that do not start on a 16-byte boundary (the vectorised histogram pass falls back to key-by-key loads)
    a = rng.integers(0, 5000, 40003).astype(np.uint16); d = torch.from_numpy(a).cuda()
    for off in (1, 3, 8):
        assert pre.median_device(d[off:]) == float(np.median(a[off:]))
    f = rng.normal(0, 100, 9001).astype(np.float32); d = torch.from_numpy(f).cuda()
    assert pre.median_device(d[1:]) == float(np.median(f[1:].astype(np.float64)))


@pytest.mark.gpu
def test_device_normalize_image_against_oracle():
    stack, _ = synth.make_stack((128, 96, 12), 40, seed=3)
    got = pre._normalize_image(stack, 100.0)
    want = pr.normalize_image(stack, 100.0)
    assert got.shape == stack.shape and got.dtype == np.float32
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-4)
    got2 = pre.lcn_gpu(stack.astype(np.float64), noise_level=5, filter_size=(27, 27, 1))
    np.testing.assert_allclose(got2, pr.lcn(stack.astype(np.float64), 5, (27, 27, 1), "constant"), rtol=0, atol=3e-4)


@pytest.mark.gpu
def test_device_normalize_full_size_properties():
    """512x512x32 uint16: linearity-type properties instead of a slow CPU run: adding a constant to the image leaves the
    result unchanged (median subtraction) wherever nothing is clamped; output is finite."""
    import torch
    stack, _ = synth.make_stack((512, 512, 32), 600, seed=0)
    d = torch.from_numpy(stack).cuda()
    a = pre.normalize_image_device(d, 100.0)
    b = pre.normalize_image_device((d.to(torch.int32) + 1000).to(torch.uint16), 100.0)
    assert bool(torch.isfinite(a).all()) and tuple(a.shape) == (512, 512, 32)
    assert torch.equal(a, b)
    with pytest.raises(ValueError):
        pre.normalize_image_device(d, 100.0, filter_size=(26, 27, 1))


@pytest.mark.gpu
@pytest.mark.parametrize("fs", [(1, 1, 1), (3, 1, 1), (1, 1, 5), (9, 3, 3), (1, 27, 1)])
def test_device_lcn_filter_shapes_against_oracle(fs):
    """Every pass structure of ct_normalize_image: no pass at all (1x1x1: source and epilogue meet in one width-1 pass), a single
    short pass, z only, three passes, y only -- zero and reflect borders, float32 images and uint16 images with the median."""
    rng = np.random.default_rng(11)
    img = (rng.normal(200.0, 30.0, (37, 29, 9)) + 500.0 * (rng.uniform(size=(37, 29, 9)) > 0.97)).astype(np.float32)
    for mode, fn in (("constant", pre.lcn_gpu), ("reflect", pre.lcn_cpu)):
        got = fn(img, noise_level=7.0, filter_size=fs)
        want = pr.lcn(img.astype(np.float64), 7.0, fs, mode)
        np.testing.assert_allclose(got, want, rtol=0, atol=3e-4)
    if fs[2] == 1:
        import torch
        raw = np.clip(img, 0, 65535).astype(np.uint16)
        got = pre.normalize_image_device(torch.from_numpy(raw).cuda(), 7.0, filter_size=fs).cpu().numpy()
        x = np.maximum(raw.astype(np.float64) - np.median(raw), 0.0)
        np.testing.assert_allclose(got, pr.lcn(x, 7.0, fs, "constant"), rtol=0, atol=3e-4)


@pytest.mark.gpu
def test_device_lcn_plane_pass_is_bit_identical_to_the_line_walk():
    """The LDS-plane box pass (default) forms exactly the sums of the per-thread line walk it replaced (CT_LCN_PLANE=0, read once per
    process: the line walk runs in its own interpreter): same bits on an odd-sized stack (partial z chunk, ragged last segment), on a
    stack whose planes need several z chunks, and on the 512-long lines of the headline frame -- both border modes."""
    import os
    import subprocess
    import sys
    import hashlib
    from pathlib import Path
    code = """
import sys, importlib, hashlib, numpy as np, torch
sys.path.insert(0, sys.argv[1])
pre = importlib.import_module("3deecelltracker_amd.preprocess")
rng = np.random.default_rng(5)
h = hashlib.sha256()
for shape in ((70, 45, 9), (300, 200, 70), (512, 512, 4)):
    raw = rng.integers(90, 4000, shape).astype(np.uint16)
    h.update(pre.normalize_image_device(torch.from_numpy(raw).cuda(), 50.0).cpu().numpy().tobytes())
    img = raw.astype(np.float32)
    h.update(np.ascontiguousarray(pre.lcn_gpu(img, noise_level=5.0, filter_size=(27, 27, 1))).tobytes())
    h.update(np.ascontiguousarray(pre.lcn_cpu(img, noise_level=5.0, filter_size=(9, 27, 3))).tobytes())
print("digest", h.hexdigest())
"""
    repo = str(Path(__file__).resolve().parents[1])
    outs = []
    for env in ({}, {"CT_LCN_PLANE": "0"}):
        r = subprocess.run([sys.executable, "-c", code, repo], capture_output=True, text=True, timeout=300, env=dict(os.environ, **env))
        assert r.returncode == 0 and "digest" in r.stdout, r.stdout[-300:] + r.stderr[-800:]
        outs.append(r.stdout.strip().splitlines()[-1])
    assert outs[0] == outs[1]
