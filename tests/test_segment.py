"""Prob map -> regions -> centres: oracle known answers (CPU), HIP vs oracle bit-exact (GPU)."""
import importlib

import numpy as np
import pytest

from oracle import segment_ref as sr

synth = importlib.import_module("3deecelltracker_amd.synth")


# ----------------------------------------------------------------------------------------------- CPU: the oracle itself
def test_oracle_known_answer_two_boxes():
    p = np.zeros((8, 7, 5), dtype=np.float32)
    p[1:3, 1:4, 1:3] = 0.9                  # 2 x 3 x 2 = 12 voxels, centre (1.5, 2, 1.5)
    p[5:8, 4:6, 0:5] = 0.6                  # 3 x 2 x 5 = 30 voxels, centre (6, 4.5, 2)
    p[4, 0, 0] = 0.8                        # speckle
    p[0, 6, 4] = 0.5                        # exactly at the threshold: background (prob > 0.5)
    lab, cen, siz = sr.segment_centroids(p, 0.5, 1, 0)
    assert siz.tolist() == [12, 1, 30]      # raster order of first voxels: (1,1,1) < (4,0,0) < (5,4,0)
    assert np.array_equal(cen, [[1.5, 2.0, 1.5], [4.0, 0.0, 0.0], [6.0, 4.5, 2.0]])
    lab, cen, siz = sr.segment_centroids(p, 0.5, 1, 2)
    assert siz.tolist() == [12, 30] and lab.max() == 2 and lab[5, 4, 0] == 2 and lab[4, 0, 0] == 0


def test_oracle_connectivity():
    p = np.zeros((4, 4, 4), dtype=np.float32)
    p[0, 0, 0] = p[1, 1, 0] = p[2, 2, 1] = 1.0      # edge neighbour, then corner neighbour
    assert sr.segment_centroids(p, 0.5, 1, 0)[2].tolist() == [1, 1, 1]
    assert sr.segment_centroids(p, 0.5, 2, 0)[2].tolist() == [2, 1]
    assert sr.segment_centroids(p, 0.5, 3, 0)[2].tolist() == [3]


def test_oracle_empty():
    lab, cen, siz = sr.segment_centroids(np.zeros((3, 3, 3), np.float32))
    assert lab.max() == 0 and cen.shape == (0, 3) and siz.size == 0


def test_synthetic_prob_map_has_touching_cells_and_speckle():
    p = synth.make_prob_map(3, (64, 64, 16), 40)
    _, _, siz = sr.segment_centroids(p, 0.5, 1, 0)
    assert (siz == 1).any() and (siz > 50).any() and p.dtype == np.float32 and p.max() <= 1.0


# ----------------------------------------------------------------------------------------------- GPU: HIP vs oracle
def _check(prob, thr, conn, min_size, cap=4096):
    import torch
    seg = importlib.import_module("3deecelltracker_amd.segment")
    lab_r, cen_r, siz_r = sr.segment_centroids(prob, thr, conn, min_size)
    lab, cen, siz = seg.segment_centroids_device(torch.from_numpy(prob).cuda(), thr, conn, min_size, cap=cap)
    assert np.array_equal(lab.cpu().numpy(), lab_r)                     # bit-exact label image
    assert np.array_equal(siz.cpu().numpy(), siz_r)
    assert np.array_equal(cen.cpu().numpy(), cen_r)                     # bit-exact fp64 centres
    return len(siz_r)


@pytest.mark.gpu
@pytest.mark.parametrize("conn", (1, 2, 3))
@pytest.mark.parametrize("min_size", (0, 20))
def test_gpu_ragged_volume(conn, min_size):
    assert _check(synth.make_prob_map(11, (67, 45, 13), 60, speckle=0.003), 0.5, conn, min_size) > 3


@pytest.mark.gpu
def test_gpu_known_answers_and_threshold_is_strict():
    p = np.zeros((8, 7, 5), dtype=np.float32)
    p[1:3, 1:4, 1:3] = 0.9; p[5:8, 4:6, 0:5] = 0.6; p[4, 0, 0] = 0.8; p[0, 6, 4] = 0.5
    assert _check(p, 0.5, 1, 0) == 3 and _check(p, 0.5, 1, 2) == 2 and _check(p, 0.7, 3, 0) == 2


@pytest.mark.gpu
def test_gpu_empty_full_and_single_voxel_volumes():
    assert _check(np.zeros((16, 16, 4), np.float32), 0.5, 1, 0) == 0
    assert _check(np.ones((33, 17, 9), np.float32), 0.5, 1, 0) == 1       # one region spanning everything
    assert _check(np.ones((1, 1, 1), np.float32), 0.5, 3, 0) == 1
    assert _check(np.ones((40, 40, 8), np.float32), 0.5, 1, 40 * 40 * 8 + 1) == 0   # min_size removes the only region


@pytest.mark.gpu
def test_gpu_snake_component_long_union_chains():
    """A one-voxel-wide serpentine through the whole volume: worst case for union-find path lengths."""
    p = np.zeros((48, 48, 6), dtype=np.float32)
    for x in range(0, 48, 2):
        p[x, :, 0] = 1.0
        if x + 1 < 48:
            p[x + 1, 47 if (x // 2) % 2 == 0 else 0, 0] = 1.0
    assert _check(p, 0.5, 1, 0) == 1


@pytest.mark.gpu
def test_gpu_speckle_many_labels_and_capacity_regrow():
    rng = np.random.default_rng(5)
    p = (rng.uniform(size=(96, 96, 8)) < 0.08).astype(np.float32)         # thousands of tiny regions
    n = _check(p, 0.5, 1, 0, cap=16)                                      # forces the capacity to grow
    assert n > 2000
    assert _check(p, 0.5, 3, 0, cap=16) < n


@pytest.mark.gpu
def test_gpu_unet_sized_volume_and_baseline_frame():
    assert _check(synth.make_prob_map(1, (160, 160, 16), 113), 0.5, 1, 20) > 20
    assert _check(synth.make_prob_map(2, (512, 512, 32), 600), 0.5, 3, 30) > 100


@pytest.mark.gpu
def test_gpu_host_mirror_raises_when_nothing_detected():
    seg = importlib.import_module("3deecelltracker_amd.segment")
    with pytest.raises(ValueError, match="No cell was detected"):
        seg.segment_centroids(np.zeros((1, 8, 8, 4, 1), np.float32))
    lab, cen, siz = seg.segment_centroids(synth.make_prob_map(4, (32, 32, 8), 6)[None, ..., None], min_size=5)
    assert lab.shape == (32, 32, 8) and cen.shape[1] == 3 and cen.dtype == np.float64 and len(siz) == len(cen)


@pytest.mark.gpu
def test_gpu_tracker_segment_prob_feeds_match():
    tracker = importlib.import_module("3deecelltracker_amd.tracker")
    ffn = importlib.import_module("3deecelltracker_amd.ffn")
    prob = synth.make_prob_map(7, (96, 96, 12), 50, speckle=0.0)
    model = ffn.FFN(); model.set_weights_dict(synth.make_ffn_weights(0))
    tk = tracker.Tracker.for_matching(model, siz_xyz=(96, 96, 12), z_xy_ratio=3.0)
    l_c, lab, r = tk.segment_prob(prob[None, ..., None], min_size=10)
    _, cen_r, _ = sr.segment_centroids(prob, 0.5, 1, 10)
    assert np.array_equal(l_c, cen_r) and np.array_equal(r, cen_r * [1.0, 1.0, 3.0]) and lab.max() == len(l_c)
    tk.set_volume1(r)
    tk.inject_segmentation(r)                 # set_volume1 reset nothing; segment_prob already injected once -- be explicit
    _, (bd, vol, _, pred) = tk.match(2)
    assert pred.shape == r.shape and np.isfinite(pred).all() and bd.shape == (len(r),)


@pytest.mark.gpu
def test_gpu_tracker_unet_cache_format(tmp_path):
    """_predict_cellregions writes unet_cache/t%06i.npy as float16 [1, x, y, z, 1] (tracker.py:668) and reads it back."""
    tracker = importlib.import_module("3deecelltracker_amd.tracker")
    unet3d = importlib.import_module("3deecelltracker_amd.unet3d")
    model = unet3d.unet3_a().set_weights_dict(synth.make_unet_weights("unet3_a", 0))
    raw, _ = synth.make_stack((64, 64, 16), 12, 0)
    tk = tracker.Tracker(volume_num=5, siz_xyz=(64, 64, 16), z_xy_ratio=2.0, z_scaling=1, noise_level=20, min_size=10, beta_tk=300,
                         lambda_tk=0.1, maxiter_tk=20, folder_path=str(tmp_path), image_name="img_t%04i_z%04i.tif",
                         unet_model_file="unet.npz", ffn_model_file="ffn.npz")
    tk.unet_model = model
    a = tk._predict_cellregions(raw, 3)
    f = tmp_path / "unet_cache" / "t000003.npy"
    cached = np.load(f)
    assert cached.dtype == np.float16 and cached.shape == (1, 64, 64, 16, 1) and a.shape == cached.shape
    assert np.array_equal(cached, np.asarray(a, dtype=np.float16))
    b = tk._predict_cellregions(None, 3)                      # served from the cache: the image is not touched
    assert b.dtype == np.float16 and np.array_equal(b, cached)
    tk.unet_model = None
    with pytest.raises(ValueError, match="load_unet"):
        tk._predict_cellregions(raw, 1)
