"""Marker-watershed region step (reference watershed.py:16-108 through Tracker._watershed): oracle properties on the CPU, device vs oracle
on the GPU.  The oracle restates four scikit-image functions (pinned against scikit-image 0.18.3 itself in tests/test_watershed_pin.py); scipy's
functions are the reference's own."""
import importlib
import sys
from pathlib import Path

import numpy as np
import pytest
import scipy.ndimage as ndi

from oracle import watershed_ref as wr

REPO = Path(__file__).resolve().parent.parent
synth = importlib.import_module("3deecelltracker_amd.synth")
seg = importlib.import_module("3deecelltracker_amd.segment")


from _ws_cases import blobs, blobs_local, random_case, random_case_large, tie_case, touching_case  # noqa: E402,F401


# ------------------------------------------------------------------------------------------------ CPU: the oracle itself
def test_gaussian_weights_are_scipys():
    from scipy.ndimage import _filters
    for sigma in (2.0, 0.3, 1.0):
        w, r = seg.gaussian_weights(sigma)
        assert np.array_equal(w, _filters._gaussian_kernel1d(sigma, 0, r)[::-1])


def test_oracle_splits_touching_cells_and_keeps_reference_conventions():
    prob = touching_case()
    labels, centres, min_size, cell_num = wr.segment_centroids(prob, 3.0, "min_size", 40)
    assert ndi.label(prob > 0.5)[1] == 3                       # connected components see the touching pair as ONE region
    assert labels.max() == 3 and cell_num == 3                 # the watershed splits it; the speck is below min_size
    assert sorted(np.round(centres[:, 0]).astype(int).tolist()) == [30, 45, 70]
    assert np.array_equal(np.unique(labels), np.arange(4))     # relabel_sequential: 1..n without gaps
    # the "cell_num" method derives min_size from the requested count (watershed.py:92)
    l2, c2, ms2, cn2 = wr.segment_centroids(prob, 3.0, "cell_num", 0, 3)
    assert cn2 == 3 and np.array_equal(l2, labels) and ms2 == np.sort(np.bincount(labels.ravel()))[0]
    with pytest.raises(ValueError):
        wr.watershed_3d(prob > 0.5, [1, 1, 3.0], "nearest", 1, 1, 3)


def test_oracle_pieces():
    # peak_local_max: plateau ties keep the smaller raveled index only; border exclusion; constant image has no peaks
    img = np.zeros((30, 30)); img[10, 10] = img[10, 12] = 2.0; img[20, 20] = 1.0; img[2, 2] = 5.0
    pk = wr.peak_local_max_mask(img, 7)
    assert sorted(zip(*np.nonzero(pk))) == [(10, 10), (20, 20)]
    assert wr.peak_local_max_mask(img, 7, exclude_border=0)[2, 2]
    assert not wr.peak_local_max_mask(np.ones((9, 9)), 2).any()
    # watershed: a 1-D ridge between two basins goes to the basin whose flood reaches it first (lower value first, then age)
    im = np.array([[0., 1., 2., 3., 2., 1., 0.]]); mk = np.zeros((1, 7), np.int32); mk[0, 0] = 1; mk[0, 6] = 2
    assert wr.watershed(im, mk, np.ones((1, 7), bool)).tolist() == [[1, 1, 1, 1, 2, 2, 2]]
    # find_boundaries(outer): pixels where two different labels meet (both sides), nothing on an object's edge to the background
    lab = np.zeros((6, 6), np.int32); lab[1:5, 1:3] = 1; lab[1:5, 3:5] = 2
    bd = wr.find_boundaries_outer(lab, 2)
    assert bd[2, 2] and bd[2, 3] and not bd[2, 1] and not bd[2, 4] and bd[0, 2]


@pytest.fixture(scope="module")
def gw(golden_dir):
    return np.load(golden_dir / "watershed.npz")


@pytest.mark.parametrize("ci", (0, 1, 2))
def test_oracle_composite_equals_the_references_own_watershed_py(gw, ci):
    """tests/golden/watershed.npz holds what the REFERENCE's watershed_2d / watershed_3d / Tracker._watershed return when scikit-image's
    four primitives are bound to the restatements (make_golden.py gen_watershed): the oracle's composite -- slice loop, boundary removal,
    sampling, min_size / cell_num bookkeeping, relabelling -- must reproduce them exactly."""
    prob = gw[f"ws_prob_{ci}"]
    zr, ms, ms_out, cn_out, ms2, cn2, trk_ms, trk_cn = gw[f"ws_para_{ci}"]
    wo, bd = wr.watershed_2d(prob, prob.shape[2], 7)
    assert np.array_equal(np.packbits(wo), gw[f"ws_wo2d_{ci}"]) and np.array_equal(np.packbits(bd), gw[f"ws_bd2d_{ci}"])
    wo_bd, clear, m1, c1 = wr.watershed_3d(wo, [1, 1, zr], "min_size", int(ms), 0, 3)
    assert (m1, c1) == (int(ms_out), int(cn_out))
    assert np.array_equal(wo_bd, gw[f"ws_wo_bd_{ci}"]) and np.array_equal(clear, gw[f"ws_clear_{ci}"])
    _, clear2, m2, c2 = wr.watershed_3d(wo, [1, 1, zr], "cell_num", 0, max(int(cn_out) - 2, 1), 3)
    assert (m2, c2) == (int(ms2), int(cn2)) and np.array_equal(clear2, gw[f"ws_clear_cellnum_{ci}"])
    seg, m3, c3 = wr.tracker_watershed(prob, zr, "min_size", int(ms), 0)
    assert (m3, c3) == (int(trk_ms), int(trk_cn)) and np.array_equal(seg, gw[f"ws_seg_auto_{ci}"])


# ------------------------------------------------------------------------------------------------ GPU: device vs oracle
@pytest.mark.gpu
@pytest.mark.parametrize("ci", (0, 1, 2))
def test_device_watershed_equals_the_reference_golden(gw, ci):
    """segmentation_auto, min_size and cell_num of Tracker._watershed as recorded from the reference's code (restated primitives)."""
    prob = gw[f"ws_prob_{ci}"]
    zr, ms, _, _, _, _, trk_ms, trk_cn = gw[f"ws_para_{ci}"]
    labels, centres, got_ms, got_cn = seg.watershed_centroids(prob, float(zr), "min_size", int(ms), 0)
    assert (got_ms, got_cn) == (int(trk_ms), int(trk_cn))
    assert np.array_equal(labels, gw[f"ws_seg_auto_{ci}"])
    n = int(labels.max())
    assert np.array_equal(centres, np.asarray(ndi.center_of_mass(labels > 0, labels, range(1, n + 1))))     # tracker.py:646-647


@pytest.mark.gpu
def test_device_edt_and_gaussian_are_scipys(gw):
    """The two scipy steps are the reference's own functions and run here: the device's per-slice EDT and its Gaussian-smoothed EDT equal
    distance_transform_edt / gaussian_filter(., 2, mode='constant') bit for bit; the 3-D stage's smoothed anisotropic EDT likewise."""
    import torch
    prob = random_case((120, 100, 16), 40, 1)
    d = torch.from_numpy(prob).cuda()
    st = seg.watershed_stages_device(d, 4.0, "2d")
    for z in range(prob.shape[2]):
        dist = ndi.distance_transform_edt(prob[:, :, z] > 0.5, sampling=[1, 1])
        assert np.array_equal(st["edt"][:, :, z], dist), z
        assert np.array_equal(st["smooth"][:, :, z], ndi.gaussian_filter(dist, 2, mode="constant")), z
        assert np.array_equal(st["window_max"][:, :, z], ndi.maximum_filter(st["smooth"][:, :, z], footprint=np.ones((15, 15), bool), mode="constant"))
    wo = st["mask_wo_boundaries"].astype(bool)
    assert np.array_equal(wo, wr.watershed_2d(prob, prob.shape[2], 7)[0])
    st3 = seg.watershed_stages_device(d, 4.0, "3d")
    dist3 = ndi.distance_transform_edt(wo, sampling=[1, 1, 4.0])
    assert np.array_equal(st3["smooth"], ndi.gaussian_filter(dist3, (2, 2, 0.3), mode="constant"))

def _device(prob, z_ratio, method, min_size, cell_num):
    labels, centres, ms, cn = seg.watershed_centroids(prob, z_ratio, method, min_size, cell_num)
    return labels, centres, ms, cn


def test_oracle_tie_rules_are_exercised():
    """The crafted tie case really has equal peak candidates (otherwise the device test below proves nothing about them)."""
    prob = tie_case()
    bn = prob[:, :, 2] > 0.5
    sm = ndi.gaussian_filter(ndi.distance_transform_edt(bn), 2, mode="constant")
    mx = ndi.maximum_filter(sm, footprint=np.ones((15, 15), bool), mode="constant")
    cand = (sm == mx) & (sm > sm.min())
    vals = sm[cand]
    assert cand.sum() > wr.peak_local_max_mask(sm, 7, exclude_border=0).sum() >= 2      # ensure_spacing dropped equal neighbours
    assert len(np.unique(vals)) < len(vals)                                              # exact ties among the candidates


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["touching", "random_small", "ragged", "empty_slices", "ties"])
def test_device_watershed_equals_the_oracle(case):
    if case == "touching":
        prob, zr, ms = touching_case(), 3.0, 40
    elif case == "random_small":
        prob, zr, ms = random_case((120, 100, 16), 40, 1), 4.0, 20
    elif case == "ragged":
        prob, zr, ms = random_case((97, 131, 21), 60, 2), 2.5, 15
    elif case == "ties":
        prob, zr, ms = tie_case(), 2.0, 5
    else:
        prob = random_case((80, 80, 20), 12, 3); prob[:, :, :6] = 0; prob[:, :, 15:] = 0; zr, ms = 5.0, 10
    want_l, want_c, want_ms, want_cn = wr.segment_centroids(prob, zr, "min_size", ms)
    got_l, got_c, got_ms, got_cn = _device(prob, zr, "min_size", ms, 0)
    assert want_l.max() >= (3 if case in ("touching", "ties") else 8)
    assert (got_ms, got_cn) == (want_ms, want_cn)
    assert np.array_equal(got_l, want_l), f"{int((got_l != want_l).sum())} voxels differ"
    assert np.array_equal(got_c, want_c)                       # integer coordinate sums / counts: bit-exact
    # the other method on the same map
    want2 = wr.segment_centroids(prob, zr, "cell_num", 0, max(want_cn - 2, 1))
    got2 = _device(prob, zr, "cell_num", 0, max(want_cn - 2, 1))
    assert (got2[2], got2[3]) == (want2[2], want2[3]) and np.array_equal(got2[0], want2[0])


@pytest.mark.gpu
def test_device_watershed_headline_size():
    """512x512x32 / ~600 cells (the benchmark's stack): labels, sizes and centres equal the oracle's."""
    stack, _ = synth.make_stack((512, 512, 32), 600, seed=0)
    prob = np.clip((stack.astype(np.float32) - 100.0) / 600.0, 0, 1)
    want_l, want_c, want_ms, want_cn = wr.segment_centroids(prob, 4.0, "min_size", 20)
    got_l, got_c, got_ms, got_cn = _device(prob, 4.0, "min_size", 20, 0)
    assert want_cn > 500 and (got_ms, got_cn) == (want_ms, want_cn)
    assert np.array_equal(got_l, want_l), f"{int((got_l != want_l).sum())} voxels differ"
    assert np.array_equal(got_c, want_c)
    assert want_cn > ndi.label(prob > 0.5)[1]                   # more cells than connected components: touching cells were split


def _hole_case():
    """two discs over four slices, the first with a one-voxel hole through its centre: the blurred EDT peaks ON the hole, i.e. a marker on a
    background voxel -- skimage's watershed drops it, its number stays used, and the first disc (no other marker) stays unlabelled"""
    prob = np.zeros((64, 64, 8), np.float32)
    g = np.stack(np.meshgrid(np.arange(64), np.arange(64), indexing="ij"), -1).astype(float)
    for z in range(2, 6):
        prob[:, :, z][((g - np.array([20, 20])) ** 2).sum(-1) <= 30] = 0.9
        prob[20, 20, z] = 0.0
        prob[:, :, z][((g - np.array([44, 44])) ** 2).sum(-1) <= 40] = 0.9
    return prob


def test_oracle_marker_on_background_leaves_an_empty_bin():
    prob = _hole_case()
    col = []
    wo, _ = wr.watershed_2d(prob, 8)
    wr.watershed_3d(wo, [1, 1, 2.0], "min_size", 0, 0, 3, collect=col)
    assert (col[0]["peaks"] & ~wo).sum() == 1 and col[0]["peaks"].sum() == 2
    labels, centres, ms, cn = wr.segment_centroids(prob, 2.0, "min_size", 0)
    assert labels.max() == 1 and (ms, cn) == (0, 2)               # np.bincount -> [background, 0, n]: three bins >= 0, minus one
    with pytest.raises(IndexError):
        wr.segment_centroids(prob, 2.0, "cell_num", 0, 3)         # np.sort(counts)[-4] of three bins (watershed.py:92)


@pytest.mark.gpu
def test_device_marker_on_background_and_min_size_zero():
    """(advisor, round 3) a marker dropped outside the mask leaves an EMPTY bin: relabel_sequential numbers present labels only, np.bincount
    has no bins past the largest present label, and cell_num beyond the bins is the reference's IndexError."""
    prob = _hole_case()
    for method, ms, cn in (("min_size", 0, 0), ("min_size", 5, 0), ("cell_num", 0, 1), ("cell_num", 0, 2)):
        want = wr.segment_centroids(prob, 2.0, method, ms, cn)
        got = _device(prob, 2.0, method, ms, cn)
        assert (got[2], got[3]) == (want[2], want[3]), (method, ms, cn)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]) and np.isfinite(got[1]).all()
    with pytest.raises(IndexError):
        _device(prob, 2.0, "cell_num", 0, 3)


@pytest.mark.gpu
def test_device_peak_tables_grow_with_the_stack():
    """More peak candidates than the first attempt's tables hold (a lattice of isolated 3 x 3 specks: 2304 per slice, 9216 in the volume --
    single-marker components, nothing to flood; 6 voxels apart, i.e. equal candidates closer than the 2-D stage's min_distance: numpy's
    introsort order decides which survive).  Until round 4 this raised; now the overflow is latched on the device (the call itself waits for
    nothing), comes back with the slots the stages wanted, and the wrapper re-runs with tables of that size -- 4096 per slice through the LDS
    sort, 16384 for the volume through the global-memory one, and the region table (cap) doubled on the way: the oracle's segmentation."""
    import torch
    prob = np.zeros((288, 288, 20), np.float32)
    for dx in range(3):
        for dy in range(3):
            prob[2 + dx::6, 2 + dy::6, 1::6] = 0.9
    want = wr.segment_centroids(prob, 4.0, "min_size", 0)
    pend = seg.watershed_centroids_enqueue(torch.from_numpy(prob).cuda(), 4.0, "min_size", 0, 0)
    labels, centres, _, ms, cn = pend.result()
    assert pend.retries >= 1 and pend.peak_cap_2d >= 4096 and pend.peak_cap_3d > 8192 and want[3] > 8192
    assert (ms, cn) == (want[2], want[3])
    assert np.array_equal(labels.cpu().numpy(), want[0]) and np.array_equal(centres.cpu().numpy(), want[1])
    got = _device(touching_case(), 3.0, "min_size", 40, 0)                      # (the latch is per call)
    want = wr.segment_centroids(touching_case(), 3.0, "min_size", 40)
    assert np.array_equal(got[0], want[0])


@pytest.mark.gpu
def test_device_watershed_on_a_large_clump():
    """One connected clump of 60 overlapping blobs (> 8192 voxels, a bounding box far beyond the LDS tile): the component takes the
    binary-heap flood, its neighbours the LDS-resident one -- labels equal the oracle's."""
    rng = np.random.default_rng(9)
    c = np.stack([rng.uniform(20, 100, 60), rng.uniform(20, 100, 60), rng.uniform(4, 12, 60)], 1)
    prob = blobs((120, 120, 16), c, rng.uniform(7, 10, 60), level=0.8)
    prob += rng.uniform(0, 0.15, prob.shape).astype(np.float32) * (prob > 0)
    lab, n = ndi.label(prob > 0.5)
    assert np.bincount(lab.ravel())[1:].max() > 8192
    want = wr.segment_centroids(prob, 3.0, "min_size", 20)
    got = _device(prob, 3.0, "min_size", 20, 0)
    assert (got[2], got[3]) == (want[2], want[3]) and want[0].max() >= 10
    assert np.array_equal(got[0], want[0]), f"{int((got[0] != want[0]).sum())} voxels differ"
    assert np.array_equal(got[1], want[1])


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(8))
def test_device_watershed_crowded_volumes_equal_the_oracle(seed):
    """Crowded random volumes (cells packed closely enough that most mask components hold several markers: the batched LDS flood does nearly all
    the labelling), different shapes incl. a single slice and odd extents, the wide-front case: labels, sizes and centres equal the oracle's."""
    rng = np.random.default_rng(100 + seed)
    shapes = [(64, 64, 8), (96, 70, 12), (57, 83, 9), (120, 120, 1), (80, 80, 6), (40, 150, 10), (110, 64, 16), (72, 72, 5)]
    shape = shapes[seed]
    n = int(np.prod(shape) / 900) + 3
    lo = np.array([4, 4, 0]); hi = np.array([shape[0] - 4, shape[1] - 4, max(shape[2] - 1, 1)])
    c = rng.uniform(lo, hi, (n, 3))
    prob = blobs(shape, c, rng.uniform(4, 9, n), z_flat=float(rng.uniform(1.5, 4.0)), level=0.8)
    prob += rng.uniform(0, 0.2, shape).astype(np.float32) * (prob > 0)
    zr, ms = float(rng.uniform(1.0, 5.0)), int(rng.integers(0, 30))
    want = wr.segment_centroids(prob, zr, "min_size", ms)
    got = _device(prob, zr, "min_size", ms, 0)
    comp = ndi.label(prob > 0.5)[1]
    assert (got[2], got[3]) == (want[2], want[3])
    assert np.array_equal(got[0], want[0]), f"{int((got[0] != want[0]).sum())} voxels differ ({comp} components, {want[3]} cells)"
    assert np.array_equal(got[1], want[1])


def test_crowded_volumes_have_multi_marker_components():
    """(what the test above relies on: in these volumes the watershed finds clearly more cells than the mask has components)"""
    more = 0
    for seed in range(8):
        rng = np.random.default_rng(100 + seed)
        shapes = [(64, 64, 8), (96, 70, 12), (57, 83, 9), (120, 120, 1), (80, 80, 6), (40, 150, 10), (110, 64, 16), (72, 72, 5)]
        shape = shapes[seed]
        n = int(np.prod(shape) / 900) + 3
        lo = np.array([4, 4, 0]); hi = np.array([shape[0] - 4, shape[1] - 4, max(shape[2] - 1, 1)])
        c = rng.uniform(lo, hi, (n, 3))
        prob = blobs(shape, c, rng.uniform(4, 9, n), z_flat=float(rng.uniform(1.5, 4.0)), level=0.8)
        prob += rng.uniform(0, 0.2, shape).astype(np.float32) * (prob > 0)
        zr, ms = float(rng.uniform(1.0, 5.0)), int(rng.integers(0, 30))
        want = wr.segment_centroids(prob, zr, "min_size", ms)
        more += int(want[3] > ndi.label(prob > 0.5)[1])
    assert more >= 4, more


@pytest.mark.gpu
@pytest.mark.parametrize("shape,n,zr,ms", [((160, 160, 64), 120, 2.0, 20), ((512, 384, 8), 200, 5.0, 10), ((233, 117, 37), 90, 3.0, 15), ((96, 96, 128), 110, 1.0, 25)],
                         ids=["deep", "wide_thin", "odd", "z128"])
def test_device_watershed_other_extents(shape, n, zr, ms):
    """A deep stack, a wide thin one, odd extents and 128 slices (the largest z of rounds 1-4): labels, sizes and centres equal the oracle's."""
    prob = random_case(shape, n, seed=sum(shape), specks=False)
    want = wr.segment_centroids(prob, zr, "min_size", ms)
    got = _device(prob, zr, "min_size", ms, 0)
    assert (got[2], got[3]) == (want[2], want[3]) and want[3] > 50
    assert np.array_equal(got[0], want[0]), f"{int((got[0] != want[0]).sum())} voxels differ"
    assert np.array_equal(got[1], want[1])


_ALT_PATHS = [{"CT_WS_FLOOD": "0"}, {"CT_WS_FLOOD": "1"}, {"CT_WS_QCAP": "8"}, {"CT_WS_SLIDE": "0"}, {"CT_WS_SELECT": "0"},
              {"CT_WS_BATCH": "0"}, {"CT_WS_BATCH": "0", "CT_WS_QCAP": "8"}, {"CT_WS_FORK": "0"}, {"CT_WS_MCAP": "3"}, {"CT_WS_EARLY_FORK": "0"}]


@pytest.mark.gpu
@pytest.mark.parametrize("env", _ALT_PATHS, ids=lambda e: "-".join(f"{k}={v}" for k, v in e.items()))
def test_device_watershed_alternative_paths_give_the_same_labels(env):
    """Every switchable path of the device watershed -- binary-heap flood for all components, global-state wave flood, the LDS flood handing
    components back when its queue is (artificially) too short (batched and one pop per round), per-voxel filters, bitonic-sort peak selection,
    the one-pop-per-round LDS flood, the serial form without the helper stream, batched rounds whose member table is (artificially) three
    entries long (the threshold is tightened until the members fit) -- produces the oracle's labels
    (the switches are read once per process: each variant runs in its own interpreter)."""
    import os
    import subprocess
    code = """
import sys, importlib, numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
from _ws_cases import touching_case, random_case, tie_case, wide_front_case
from oracle import watershed_ref as wr
seg = importlib.import_module("3deecelltracker_amd.segment")
bad = 0
for prob, zr, ms in ((touching_case(), 3.0, 40), (random_case((120, 100, 16), 40, 1), 4.0, 20), (tie_case(), 2.0, 5), (wide_front_case(), 3.0, 40)):
    want = wr.segment_centroids(prob, zr, "min_size", ms)
    got = seg.watershed_centroids(prob, zr, "min_size", ms, 0)
    bad += int(not np.array_equal(got[0], want[0])) + int(not np.array_equal(got[1], want[1]))
print("mismatches", bad)
"""
    repo = str(REPO)
    r = subprocess.run([sys.executable, "-c", code, repo], capture_output=True, text=True, timeout=300, env=dict(os.environ, **env))
    assert r.returncode == 0 and "mismatches 0" in r.stdout, r.stdout[-300:] + r.stderr[-800:]


@pytest.mark.gpu
def test_device_watershed_enqueued_on_another_stream_and_with_too_small_a_table():
    """watershed_centroids_enqueue: the kernels run on the stream that was current at the call, result() may be taken on another stream (and
    later); a centre table that turns out too small (cap < regions) is grown and the call repeated inside result().  Same regions either way."""
    import torch
    from _ws_cases import random_case
    prob = torch.from_numpy(random_case((120, 100, 16), 40, 1)).cuda()
    want = seg.watershed_centroids_device(prob, 4.0, "min_size", 20)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        pend = seg.watershed_centroids_enqueue(prob, 4.0, "min_size", 20, cap=8)      # fewer slots than regions: regrown in result()
        pend2 = seg.watershed_centroids_enqueue(prob, 4.0, "min_size", 20, want_labels=False)
    got, got2 = pend.result(), pend2.result()
    assert len(want[1]) > 8 and pend.cap >= len(want[1])
    assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1]) and torch.equal(got[2], want[2]) and got[3:] == want[3:]
    assert got2[0] is None and torch.equal(got2[1], want[1])


# ------------------------------------------------------------------------------------------------ no table limits (round 5)
def _device_caps(prob, z_ratio, min_size, p2, p3):
    import torch
    d = torch.from_numpy(np.ascontiguousarray(prob, dtype=np.float32)).cuda()
    pend = seg.watershed_centroids_enqueue(d, z_ratio, "min_size", min_size, 0, peak_cap_2d=p2, peak_cap_3d=p3)
    labels, centres, _, ms, cn = pend.result()
    return labels.cpu().numpy(), centres.cpu().numpy(), ms, cn, pend


@pytest.mark.gpu
def test_device_watershed_takes_more_than_128_slices():
    """The reference's watershed.py takes any stack; until round 4 the device refused z > 128 (per-slice statistics tables of 128 entries).
    A 150- and a 300-slice stack: labels, sizes and centres equal the oracle's."""
    for shape, n, zr in (((72, 64, 150), 90, 1.0), ((40, 48, 300), 70, 2.0)):
        prob = random_case(shape, n, seed=sum(shape), specks=False)
        want = wr.segment_centroids(prob, zr, "min_size", 15)
        got = _device(prob, zr, "min_size", 15, 0)
        assert (got[2], got[3]) == (want[2], want[3]) and want[3] > 30
        assert np.array_equal(got[0], want[0]), f"{int((got[0] != want[0]).sum())} voxels differ"
        assert np.array_equal(got[1], want[1])


@pytest.mark.gpu
def test_device_watershed_grows_its_peak_tables():
    """Peak-candidate tables that are too small for the stack (16 slots per slice / in the volume, where the stack wants ~10 / ~60) overflow on the
    device, the overflow comes back with the slots the stages wanted, and the wrapper re-runs with tables of that size: same result as with
    the default tables.  Both stages overflow, so the call takes two extra rounds."""
    prob = random_case((120, 100, 16), 60, 13)
    want = wr.segment_centroids(prob, 3.0, "min_size", 20)
    ref = _device_caps(prob, 3.0, 20, seg.PEAK_CAP_2D, seg.PEAK_CAP_3D)
    assert ref[4].retries == 0 and np.array_equal(ref[0], want[0])
    got = _device_caps(prob, 3.0, 20, 16, 16)
    assert got[4].retries >= 1 and got[4].peak_cap_2d > 16 and got[4].peak_cap_3d > 16
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]) and got[2:4] == (want[2], want[3])
    # only the volume's table too small: one retry, the 2-D tables stay
    got3 = _device_caps(prob, 3.0, 20, seg.PEAK_CAP_2D, 32)
    assert got3[4].retries == 1 and got3[4].peak_cap_2d == seg.PEAK_CAP_2D and np.array_equal(got3[0], want[0])


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["ties", "random_c", "clean_c"])
def test_device_watershed_global_memory_selection_equals_the_lds_forms(case):
    """Tables beyond what LDS sorts (8192 candidates per group): ws_peak_select_kernel<GLOBAL> runs the same selection on a scratch slab of the
    workspace -- incl. numpy's introsort among exactly tied candidates (the designed tie volume) as a one-thread replay with numpy's own stack.
    CT_WS_SELECT=0 sends every group through that kernel (the counting form takes groups of <= 2048 otherwise): run in its own interpreter."""
    import os
    import subprocess
    code = """
import sys, importlib, numpy as np, torch
sys.path.insert(0, %r); sys.path.insert(0, %r)
from _ws_cases import PIN_CASES
from oracle import watershed_ref as wr
seg = importlib.import_module("3deecelltracker_amd.segment")
synth = importlib.import_module("3deecelltracker_amd.synth")
build, zr, ms = PIN_CASES[%r]
prob = build(synth.make_stack)
want = wr.segment_centroids(prob, zr, "min_size", ms)
d = torch.from_numpy(np.ascontiguousarray(prob, dtype=np.float32)).cuda()
for p2, p3 in ((16384, 16384), (2048, 32768)):
    lab, cen, _, m, c = seg.watershed_centroids_device(d, zr, "min_size", ms, 0, peak_cap_2d=p2, peak_cap_3d=p3)
    assert (m, c) == (want[2], want[3]), (m, c, want[2], want[3])
    assert np.array_equal(lab.cpu().numpy(), want[0]), int((lab.cpu().numpy() != want[0]).sum())
    assert np.array_equal(cen.cpu().numpy(), want[1])
print("global selection ok")
""" % (str(REPO), str(REPO / "tests"), case)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, CT_WS_SELECT="0"), cwd=REPO)
    assert r.returncode == 0 and "global selection ok" in r.stdout, r.stdout[-500:] + r.stderr[-2000:]


@pytest.mark.gpu
@pytest.mark.parametrize("shape,n,zr", [((512, 1024, 21), 2000, 5.0), ((299, 499, 98), 2000, 1.5)], ids=["21x512x1024", "98x299x499"])
def test_device_watershed_notebook_shapes_with_2000_cells(shape, n, zr):
    """The stack shapes of the reference's own notebooks (track_stardist_single_mode.ipynb: 21 x 512 x 1024; -h5.ipynb: 98 x 299 x 499, given
    there as z, x, y) with ~2000 blobs: labels, sizes and centres equal the oracle's, with the default tables (no retry needed)."""
    prob = random_case_large(shape, n, seed=shape[2])
    want = wr.segment_centroids(prob, zr, "min_size", 20)
    got = _device_caps(prob, zr, 20, seg.PEAK_CAP_2D, seg.PEAK_CAP_3D)
    assert got[4].retries == 0 and got[2:4] == (want[2], want[3]) and want[3] > 1500
    assert np.array_equal(got[0], want[0]), f"{int((got[0] != want[0]).sum())} voxels differ"
    assert np.array_equal(got[1], want[1])
