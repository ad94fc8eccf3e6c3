"""The marker watershed held to THE REFERENCE'S OWN CODE RUN ON THE REAL scikit-image 0.18.3 (tests/golden/watershed_skimage.npz, recorded
by tests/golden/make_watershed_golden.py under the image's second interpreter, /opt/conda/bin/python3.9: CellTracker/watershed.py and
Tracker._watershed unmodified, nothing of scikit-image / scipy stubbed).  CPU: the oracle's restated primitives and its composite equal
every recorded stage incl. the choice among exactly tied peak candidates (numpy's generic introsort, restated).  GPU: the device path equals
the recorded segmentation, sizes bookkeeping and centres on eight volumes incl. the 512 x 512 x 32 benchmark stack -- compared with the
golden directly, no oracle in between."""
import importlib

import numpy as np
import pytest

import _ws_cases as cases
from oracle import watershed_ref as wr

synth = importlib.import_module("3deecelltracker_amd.synth")


@pytest.fixture(scope="module")
def pin(golden_dir):
    return np.load(golden_dir / "watershed_skimage.npz")


def _case(pin, name):
    build, zr, ms = cases.PIN_CASES[name]
    prob = build(synth.make_stack)
    assert cases.sha(prob) == str(pin[f"{name}_sha"]), "the regenerated input differs from the one the golden was recorded on"
    assert (zr, ms) == (float(pin[f"{name}_para"][0]), int(pin[f"{name}_para"][1]))
    return prob, zr, ms


def test_golden_is_the_real_thing(pin):
    v = [str(s) for s in pin["versions"]]
    assert v[0].startswith("scikit-image 0.1") and [str(s) for s in pin["names"]] == list(cases.PIN_CASES)
    assert int(pin["headline_para"][3]) == 566 and int(pin["headline_seg_auto"].max()) == 566


def _tied_marker_components(mask, peaks, smooth):
    """mask components (full connectivity, as the flood sees one basin system: connectivity 1) that hold two or more markers of EXACTLY equal
    height: there the order in which upstream's binary heap pops the equal (value, age 0) seeds decides the boundary between them"""
    import scipy.ndimage as ndi
    comp, n = ndi.label(mask)
    out = np.zeros(mask.shape, bool)
    for c in range(1, n + 1):
        v = smooth[(comp == c) & peaks]
        if v.size > 1 and np.unique(v).size < v.size:
            out |= comp == c
    return out


ALL_SMALL = cases.TIE_FREE + ("random_a", "random_b", "random_c", "ties")


@pytest.mark.parametrize("name", ALL_SMALL)
def test_restated_primitives_equal_skimage_stage_by_stage(pin, name):
    """Every restated primitive against scikit-image's own output, each fed with the RECORDED output of the stage before it, EXACTLY, on all
    eight small volumes incl. the designed tie volume: peak_local_max (the choice among exactly tied candidates: numpy's introsort restated,
    oracle.argsort_quicksort), label + watershed (seeds of exactly equal height in upstream's heap order, oracle._UpstreamHeap),
    find_boundaries, the 2-D composite, label + watershed in 3-D, the size bookkeeping."""
    import scipy.ndimage as ndi
    prob, zr, ms = _case(pin, name)
    unpack = lambda key, dt=bool: np.unpackbits(pin[f"{name}_{key}"])[:prob.size].reshape(prob.shape).astype(dt)
    peaks2d, bd2d, wo2d, peaks3d = unpack("peaks2d"), unpack("bd2d"), unpack("wo2d"), unpack("peaks3d")
    labels2d = pin[f"{name}_labels2d"]
    for z in range(prob.shape[2]):
        bn = prob[:, :, z] > 0.5
        smooth = ndi.gaussian_filter(ndi.distance_transform_edt(bn, sampling=[1, 1]), 2, mode="constant")
        assert np.array_equal(wr.peak_local_max_mask(smooth, 7), peaks2d[:, :, z]), f"peak_local_max, slice {z}"
        lab = wr.watershed(-smooth, wr.label_full(peaks2d[:, :, z]), bn, seed_order="upstream")
        assert np.array_equal(lab, labels2d[:, :, z]), f"label + watershed, slice {z}"
        assert np.array_equal(wr.find_boundaries_outer(lab, 2), bd2d[:, :, z]), f"find_boundaries, slice {z}"
    wo = prob > 0.5
    wo[bd2d] = False
    assert np.array_equal(wo, wo2d)
    smooth3 = ndi.gaussian_filter(ndi.distance_transform_edt(wo2d, sampling=[1, 1, zr]), (2, 2, 0.3), mode="constant")
    assert np.array_equal(wr.peak_local_max_mask(smooth3, 3, exclude_border=0), peaks3d), "peak_local_max (3-D)"
    lab3 = wr.watershed(-smooth3, wr.label_full(peaks3d), wo2d, seed_order="upstream")
    assert np.array_equal(lab3, pin[f"{name}_labels3d"]), "label + watershed (3-D)"
    counts = np.sort(np.bincount(lab3.ravel()))
    assert int(np.sum(counts >= ms) - 1) == int(pin[f"{name}_para"][3])
    assert np.array_equal(wr.relabel_sequential(wr.remove_small_objects(lab3, ms)), pin[f"{name}_seg_auto"])


@pytest.mark.parametrize("name", ALL_SMALL)
def test_seed_order_is_the_only_open_rule_and_only_bites_on_the_tie_volume(pin, name):
    """Upstream pops seeds of exactly equal height in the order ONE heap over the whole image leaves them in (the oracle's default, and the
    device's rule since round 4); until round 3 the device took the smaller raveled index (seed_order="raveled").  Whole pipeline, both
    rules: with upstream's the oracle equals the reference on every volume; with the raveled rule it does on all but the designed tie
    volume -- and there only inside basin systems that hold two seeds of exactly equal height."""
    import scipy.ndimage as ndi
    prob, zr, ms = _case(pin, name)
    want = pin[f"{name}_seg_auto"]
    up, ucen, ums, ucn = wr.segment_centroids(prob, zr, "min_size", ms)
    assert np.array_equal(up, want) and (ums, ucn) == (int(pin[f"{name}_para"][2]), int(pin[f"{name}_para"][3]))
    assert np.array_equal(ucen, pin[f"{name}_centres"])
    own, _, _, _ = wr.segment_centroids(prob, zr, "min_size", ms, seed_order="raveled")
    if name != "ties":
        assert np.array_equal(own, want)
        return
    assert not np.array_equal(own, want)
    # 2-D stage: every differing pixel lies in a basin system with tied seeds
    peaks2d = np.unpackbits(pin["ties_peaks2d"])[:prob.size].reshape(prob.shape).astype(bool)
    for z in range(prob.shape[2]):
        bn = prob[:, :, z] > 0.5
        smooth = ndi.gaussian_filter(ndi.distance_transform_edt(bn, sampling=[1, 1]), 2, mode="constant")
        a = wr.watershed(-smooth, wr.label_full(peaks2d[:, :, z]), bn, seed_order="raveled")
        tied = _tied_marker_components(bn, peaks2d[:, :, z], smooth)
        assert np.array_equal(a[~tied], pin["ties_labels2d"][:, :, z][~tied])


@pytest.mark.parametrize("name", cases.TIE_FREE)
def test_cell_num_method_equals_the_reference(pin, name):
    prob, zr, ms = _case(pin, name)
    wo = np.unpackbits(pin[f"{name}_wo2d"])[:prob.size].reshape(prob.shape).astype(bool)
    cn_in, ms2, cn2 = (int(v) for v in pin[f"{name}_cellnum"])
    _, clear_cn, oms2, ocn2 = wr.watershed_3d(wo, [1, 1, zr], "cell_num", 0, cn_in, 3)
    assert (oms2, ocn2) == (ms2, cn2) and np.array_equal(clear_cn, pin[f"{name}_clear_cellnum"])


def test_benchmark_stack_equals_the_reference(pin):
    """512 x 512 x 32 / 566 cells, all 8.4 M voxels and the centres -- incl. the 45 exactly tied pairs of peak candidates (blobs symmetric about
    a pixel edge) that numpy's introsort orders (with numpy's AVX-512 argsort, >= 1.25 on such CPUs, upstream itself keeps the other pixel of
    each pair and 96 voxels come out differently: tests/golden/make_watershed_golden.py)."""
    prob, zr, ms = _case(pin, "headline")
    labels, centres, oms, ocn = wr.segment_centroids(prob, zr, "min_size", ms)
    assert (oms, ocn) == (int(pin["headline_para"][2]), 566)
    assert np.array_equal(labels, pin["headline_seg_auto"])
    assert np.array_equal(centres, pin["headline_centres"])
    n = prob.size
    p2 = np.unpackbits(pin["headline_peaks2d"])[:n].reshape(prob.shape).astype(bool)
    import scipy.ndimage as ndi
    for z in (0, 9, 17, 31):
        smooth = ndi.gaussian_filter(ndi.distance_transform_edt(prob[:, :, z] > 0.5, sampling=[1, 1]), 2, mode="constant")
        assert np.array_equal(wr.peak_local_max_mask(smooth, 7), p2[:, :, z])


def test_argsort_restatement_equals_numpy_generic_sort():
    """oracle.argsort_quicksort against np.argsort itself on the generic code path (the image's second interpreter with the AVX-512 dispatch
    switched off): heavy ties, no ties, presorted, all-equal, organ-pipe (the heapsort fallback), sizes 0 ... 5000."""
    import os
    import subprocess
    from pathlib import Path
    py = "/opt/conda/bin/python3.9"
    if not os.path.exists(py):
        pytest.skip("no second interpreter on this machine")
    code = """
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
from numpy.core._multiarray_umath import __cpu_features__ as cpu
assert not cpu['AVX512_SKX']
from oracle.watershed_ref import argsort_quicksort as aq
rng = np.random.default_rng(0); bad = 0; total = 0
for trial in range(600):
    n = int(rng.integers(0, 40)) if trial % 3 == 0 else int(rng.integers(0, 4000)) if trial % 3 == 1 else int(rng.integers(0, 300))
    k = int(rng.integers(1, 30)); mode = trial % 4
    a = (rng.integers(0, k, n).astype(float) if mode == 0 else np.round(rng.normal(size=n), 1) if mode == 1 else rng.normal(size=n) if mode == 2
         else -np.sort(rng.integers(0, k, n).astype(float)))
    total += 1; bad += not np.array_equal(np.argsort(-a), aq(-a))
for n in (16, 17, 33, 1000, 5000):
    for a in (np.zeros(n), np.arange(n) % 2.0, np.concatenate([np.arange(n // 2), np.arange(n - n // 2, 0, -1)]).astype(float)):
        total += 1; bad += not np.array_equal(np.argsort(-a), aq(-a))
print('mismatches', bad, 'of', total)
"""
    env = dict(os.environ, NPY_DISABLE_CPU_FEATURES="AVX512F AVX512CD AVX512_SKX AVX512_CLX AVX512_CNL AVX512_ICL", PYTHONDONTWRITEBYTECODE="1")
    r = subprocess.run([py, "-W", "ignore", "-c", code, str(Path(__file__).resolve().parent.parent)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "mismatches 0 of 615" in r.stdout, r.stdout[-300:] + r.stderr[-500:]


def test_upstream_heap_restatement_equals_skimage_watershed():
    """oracle.watershed(seed_order="upstream") against skimage.segmentation.watershed itself (second interpreter) on inputs made of ties:
    integer-valued images with 2-12 random seeds, random masks, 2-D and 3-D."""
    import os
    import subprocess
    from pathlib import Path
    py = "/opt/conda/bin/python3.9"
    if not os.path.exists(py):
        pytest.skip("no second interpreter on this machine")
    code = """
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
from skimage.segmentation import watershed as sk
from oracle.watershed_ref import watershed as ours
rng = np.random.default_rng(0); bad = 0; total = 0
for t in range(60):
    nd = 2 if t < 45 else 3
    shape = tuple(int(rng.integers(8, 40)) for _ in range(2)) if nd == 2 else (int(rng.integers(6, 14)),) * 3
    img = rng.integers(0, 4 if nd == 2 else 3, shape).astype(float)
    mask = rng.uniform(size=shape) > 0.15
    mk = np.zeros(shape, np.int32); k = int(rng.integers(2, 12))
    mk.ravel()[rng.choice(img.size, k, replace=False)] = np.arange(1, k + 1)
    total += 1; bad += not np.array_equal(sk(img, mk, mask=mask), ours(img, mk, mask, seed_order='upstream'))
print('mismatches', bad, 'of', total)
"""
    r = subprocess.run([py, "-W", "ignore", "-c", code, str(Path(__file__).resolve().parent.parent)], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1"))
    assert r.returncode == 0 and "mismatches 0 of 60" in r.stdout, r.stdout[-300:] + r.stderr[-500:]


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(cases.PIN_CASES))
def test_device_watershed_equals_the_reference_on_real_skimage(pin, name):
    """All nine recorded volumes, the designed tie volume included: there two equal seeds share a mask component, the groups concerned are
    replayed with upstream's heap on the device (ws_flood_upstream_kernel), and labels, sizes, bookkeeping and centres equal what the
    reference produced on the real scikit-image -- compared with the golden directly, no oracle in between."""
    seg = importlib.import_module("3deecelltracker_amd.segment")
    import torch
    prob, zr, ms = _case(pin, name)
    labels, centres, sizes, oms, ocn = seg.watershed_centroids_device(torch.from_numpy(prob).cuda(), zr, "min_size", ms, 0)
    assert (oms, ocn) == (int(pin[f"{name}_para"][2]), int(pin[f"{name}_para"][3]))
    got = labels.cpu().numpy()
    want = pin[f"{name}_seg_auto"]
    assert np.array_equal(got, want), f"{int((got != want).sum())} voxels differ"
    assert np.array_equal(centres.cpu().numpy(), pin[f"{name}_centres"])
    assert np.array_equal(sizes.cpu().numpy(), np.bincount(want.ravel())[1:])
