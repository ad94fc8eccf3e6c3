"""CPU-only: host-side mirrors (no kernels) against the golden vectors produced by the reference."""
import importlib
import json

import numpy as np
import pytest

cit = importlib.import_module("3deecelltracker_amd.coord_image_transformer")
ffn_mod = importlib.import_module("3deecelltracker_amd.ffn")
track = importlib.import_module("3deecelltracker_amd.track")
tl = importlib.import_module("3deecelltracker_amd.trackerlite")
unet3d = importlib.import_module("3deecelltracker_amd.unet3d")
synth = importlib.import_module("3deecelltracker_amd.synth")
arch_mod = importlib.import_module("3deecelltracker_amd.arch")


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(golden_dir / "match.npz")


@pytest.fixture(scope="module")
def meta(golden_dir):
    return json.loads((golden_dir / "match.json").read_text())


def test_coordinates_against_reference(g):
    cr, vs = g["coords_in"], g["coords_vs"]
    views = {"raw": cit.Coordinates(cr, 5, vs, "raw"), "real": cit.Coordinates(cr, 5, vs, "real"),
             "interp": cit.Coordinates(cr, 5, vs, "interp")}
    for tag, c in views.items():
        assert c._raw.dtype == np.float32 and c.cell_num == 9
        assert np.array_equal(c.real, g[f"coords_{tag}_real"])
        assert np.array_equal(c.interp, g[f"coords_{tag}_interp"]) and c.interp.dtype == np.int32
        assert np.array_equal(c.raw, g[f"coords_{tag}_raw"])
    assert np.array_equal((views["raw"] + views["real"]).real, g["coords_add_real"])
    assert np.array_equal((views["raw"] - views["interp"]).real, g["coords_sub_real"])
    with pytest.raises(ValueError):
        cit.Coordinates(cr, 5, vs, "voxels")


def test_normalize_points_argument_errors():
    with pytest.raises(ValueError):
        ffn_mod.normalize_points(np.zeros(5))
    with pytest.raises(ValueError):
        ffn_mod.normalize_points(np.zeros((5, 2)))


def test_ensemble_schedules_against_reference(meta):
    for c in meta["get_volumes_list"]:
        assert tl.get_volumes_list(c["cur"], c["skip"], c["samp"], c["adj"], c["start"]) == c["out"], c
    for c in meta["get_reference_vols"]:
        assert track.get_reference_vols(c["ens"], c["vol"], c["adj"]) == c["out"], c
    with pytest.raises(AssertionError):
        tl.get_volumes_list(1, [], 20, False, 1)
    assert tl.get_volumes_list(80, [79], 20) == list(range(20, 78, 3))        # SURVEY 8a a13 example


def test_get_sizes_padded_im():
    assert unet3d._get_sizes_padded_im(512, 112) == (560, 5)
    assert unet3d._get_sizes_padded_im(112, 112) == (112, 1)
    assert unet3d._get_sizes_padded_im(113, 112) == (224, 2)


def test_weight_flattening_layout():
    w = synth.make_unet_weights("unet3_a", 0)
    flat = unet3d.flatten_unet_weights(w)
    assert flat.dtype == np.float32 and flat.size == 512025
    k0 = w["convs"][0]["kernel"].ravel()
    assert np.array_equal(flat[:k0.size], k0) and np.array_equal(flat[k0.size:k0.size + 8], w["convs"][0]["bias"])
    assert np.array_equal(flat[-1:], w["head"]["bias"]) and np.array_equal(flat[-9:-1], w["head"]["kernel"].ravel())
    fw = synth.make_ffn_weights(0)
    ff = ffn_mod.flatten_ffn_weights(fw)
    assert ff.size == 61 * 512 + 4 * 512 + 1024 * 512 + 4 * 512 + 512 + 1
    assert np.array_equal(ff[:61 * 512], fw["w1"].ravel()) and ff[-1] == fw["b3"][0]


def test_model_objects_expose_keras_surface_without_gpu():
    m = unet3d.unet3_a()
    assert m.input_shape == (None, 160, 160, 16, 1) and m.output_shape == (None, 160, 160, 16, 1)
    assert unet3d.unet3_b().input_shape == (None, 96, 96, 8, 1)
    assert unet3d.unet3_c().input_shape == (None, 64, 64, 64, 1)
    with pytest.raises(NotImplementedError):
        m.fit_generator()


def test_trackerlite_constructor_errors(tmp_path, g):
    proof = cit.Coordinates(g["coords_in"], 5, g["coords_vs"], "raw")
    with pytest.raises(TypeError):
        tl.TrackerLite(str(tmp_path), "x", proof, miss_frame=(1, 2))
    with pytest.raises(ValueError):                  # missing weight file -> ValueError wrapping OSError
        tl.TrackerLite(str(tmp_path), "missing_model", proof, basedir=str(tmp_path))
    assert (tmp_path / "track_results" / "coords_real").is_dir()


def test_trained_ffn_fixture_loads_and_discriminates(golden_dir):
    """3deecelltracker_amd/data/ffn_synthetic_trained.npz (tests/golden/train_synthetic_ffn.py): layout of make_ffn_weights; through the oracle it scores a
    true pair near 1 and a wrong pair near 0."""
    import importlib
    from oracle import match_ref as mr
    synth = importlib.import_module("3deecelltracker_amd.synth")
    w = synth.load_ffn_npz(synth.TRAINED_FFN_PATH)
    ref = synth.make_ffn_weights(0)
    assert {k: (v.shape if hasattr(v, "shape") else {kk: vv.shape for kk, vv in v.items()}) for k, v in w.items()} == \
           {k: (v.shape if hasattr(v, "shape") else {kk: vv.shape for kk, vv in v.items()}) for k, v in ref.items()}
    rng = np.random.default_rng(3)
    x = mr.normalize_points(rng.uniform(0, 1, (80, 3)))
    y = x @ (np.eye(3) + (rng.uniform(0, 1, (3, 3)) - 0.5) * 0.1) + rng.normal(0, 1e-3, x.shape)
    corr = mr.initial_matching(lambda q: mr.ffn_forward(w, q), x, y, 20)
    diag = np.diag(corr); off = corr[~np.eye(80, dtype=bool)]
    assert np.median(diag) > 0.9 and np.median(off) < 0.01


def test_bench_match_schedule_never_queues_more_batches_than_chains():
    """bench.match_schedule: defaults per pipeline, and a short run (the driver's --steps 20) gets ceil(K / chains) frames per
    chain so that no batch waits behind another one at the end of the timed region."""
    import importlib.util
    from pathlib import Path
    spec = importlib.util.spec_from_file_location("bench_mod", Path(__file__).resolve().parent.parent / "bench.py")
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    assert bench.match_schedule(128, False, None, None) == (1, 32)
    assert bench.match_schedule(20, False, None, None) == (1, 20)
    assert bench.match_schedule(128, True, None, None) == (3, 16)
    assert bench.match_schedule(20, True, None, None) == (3, 7)
    assert bench.match_schedule(5, False, 2, 16) == (2, 3)
    for k in (1, 3, 20, 128):
        for w in (1, 2, 3):
            _, b = bench.match_schedule(k, False, w, 16)
            assert b >= 1 and -(-k // b) <= max(w, -(-k // 16))


def test_legacy_match_without_volume_size_flags_no_boundary_cells(monkeypatch):
    """Tracker.for_matching() has no siz_xyz: match() must leave the boundary flags alone (`a[()] = 1` flagged every cell)."""
    tracker_mod = importlib.import_module("3deecelltracker_amd.tracker")
    trk = tracker_mod.Tracker.for_matching(None)
    pts = np.random.default_rng(0).uniform(0, 50, size=(7, 3))
    trk.set_volume1(pts)
    trk._injected = True                                    # what inject_segmentation() records (it uploads to the device)
    monkeypatch.setattr(trk, "_predict_pos_once", lambda source_volume, draw: (pts + 0.25, None))
    anim, (bd, vol, i_disp, pred) = trk.match(3)
    assert anim is None and vol == 3 and i_disp is None
    assert bd.shape == (7,) and not bd.any()
    trk.cells_on_boundary[2] = 1                            # flags of earlier volumes are kept
    trk._injected = True
    assert trk.match(4)[1][0].tolist() == [0, 0, 1, 0, 0, 0, 0]


def test_transform_disps_keeps_integer_arrays_like_the_reference():
    """tracker.py:553-556 assigns `new[:, 2] = new[:, 2] * factor`: an int array (what _transform_real_to_interpolated returns and the
    reference feeds back in, :307, :451) keeps its dtype and the scaled z is truncated.  Expected values recorded from the reference's
    own Tracker._transform_disps (numpy 2.2): an in-place `*=` raises UFuncTypeError instead."""
    tracker_mod = importlib.import_module("3deecelltracker_amd.tracker")
    a = np.array([[1, 2, 3], [4, 5, -7], [0, 0, 10]])
    for factor, want in ((1.9, [[1, 2, 5], [4, 5, -13], [0, 0, 19]]), (0.2, [[1, 2, 0], [4, 5, -1], [0, 0, 2]]),
                         (5 / 1.9, [[1, 2, 7], [4, 5, -18], [0, 0, 26]])):
        got = tracker_mod.Tracker._transform_disps(a, factor)
        assert got.dtype == a.dtype and got.tolist() == want
    assert a.tolist() == [[1, 2, 3], [4, 5, -7], [0, 0, 10]]                     # a fresh array every time
    f = tracker_mod.Tracker._transform_disps(a.astype(float), 1.9)
    assert f.dtype == np.float64 and f.tolist() == [[1.0, 2.0, 5.699999999999999], [4.0, 5.0, -13.299999999999999], [0.0, 0.0, 19.0]]
    trk = tracker_mod.Tracker.for_matching(None)
    trk.z_xy_ratio, trk.z_scaling = 1.9, 5
    i_disp = trk._transform_real_to_interpolated(np.array([[1.0, 2.0, 3.3]]))    # int array ...
    assert i_disp.dtype.kind == "i"
    assert trk._transform_interpolated_to_layer(i_disp).dtype.kind == "i"        # ... fed back in, as the reference does
