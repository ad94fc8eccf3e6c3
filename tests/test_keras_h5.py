"""Keras `.h5` import: files written in the Keras 2.x HDF5 layout are read back into the weight containers.

Two back ends: the real h5py wherever it is importable (it is not part of the build image: those cases are skipped there), and
a dict-backed stand-in exposing the part of the h5py API the reader uses (File / groups / attrs / path look-ups), so that the
reader's pairing and validation logic executes in every environment."""
import importlib
import sys
import types

import numpy as np
import pytest


class _FakeGroup:
    def __init__(self):
        self.attrs = {}
        self._items = {}

    def create_group(self, name):
        g = self
        for part in name.split("/"):
            g = g._items.setdefault(part, _FakeGroup())
        return g

    def create_dataset(self, name, data):
        *parents, leaf = name.split("/")
        g = self.create_group("/".join(parents)) if parents else self
        g._items[leaf] = np.array(data)

    def __getitem__(self, name):
        g = self
        for part in name.split("/"):
            g = g._items[part]
        return g

    def __contains__(self, name):
        try:
            self[name]
            return True
        except KeyError:
            return False

    def keys(self):
        return self._items.keys()


_FAKE_FILES = {}


class _FakeFile(_FakeGroup):
    def __new__(cls, path, mode="r"):
        if mode == "w":
            obj = super().__new__(cls); _FakeGroup.__init__(obj); _FAKE_FILES[str(path)] = obj
            return obj
        return _FAKE_FILES[str(path)]

    def __init__(self, path, mode="r"):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


@pytest.fixture(params=["stand-in", "h5py"], autouse=True)
def h5py(request, monkeypatch):
    if request.param == "h5py":
        return pytest.importorskip("h5py")
    fake = types.ModuleType("h5py"); fake.File = _FakeFile
    monkeypatch.setitem(sys.modules, "h5py", fake)
    globals()["h5py"] = fake
    return fake

synth = importlib.import_module("3deecelltracker_amd.synth")
kh5 = importlib.import_module("3deecelltracker_amd.keras_h5")
ARCHS = importlib.import_module("3deecelltracker_amd.arch").ARCHS


def _write_unet(path, w, full_model, h5py=None):
    h5py = sys.modules['h5py']
    """Functional model: layers in creation order conv3d, leaky_re_lu, batch_normalization, conv3d_1, ... (unet3d.py:84-119)."""
    with h5py.File(path, "w") as f:
        root = f.create_group("model_weights") if full_model else f
        names = ["input_1"]
        root.create_group("input_1").attrs["weight_names"] = []

        def suffix(base, i):
            return base if i == 0 else f"{base}_{i}"
        for i, l in enumerate(w["convs"]):
            cn, bn = suffix("conv3d", i), suffix("batch_normalization", i)
            g = root.create_group(cn)
            g.attrs["weight_names"] = [f"{cn}/kernel:0".encode(), f"{cn}/bias:0".encode()]
            g.create_dataset(f"{cn}/kernel:0", data=l["kernel"]); g.create_dataset(f"{cn}/bias:0", data=l["bias"])
            act = suffix("leaky_re_lu", i)
            root.create_group(act).attrs["weight_names"] = []
            g = root.create_group(bn)
            keys = (("gamma", "gamma"), ("beta", "beta"), ("moving_mean", "mean"), ("moving_variance", "var"))
            g.attrs["weight_names"] = [f"{bn}/{k}:0".encode() for k, _ in keys]
            for k, mine in keys:
                g.create_dataset(f"{bn}/{k}:0", data=l[mine])
            names += [cn, act, bn]
        hn = suffix("conv3d", len(w["convs"]))
        g = root.create_group(hn)
        g.attrs["weight_names"] = [f"{hn}/kernel:0".encode(), f"{hn}/bias:0".encode()]
        g.create_dataset(f"{hn}/kernel:0", data=w["head"]["kernel"]); g.create_dataset(f"{hn}/bias:0", data=w["head"]["bias"])
        names.append(hn)
        root.attrs["layer_names"] = [n.encode() for n in names]


def _write_ffn(path, w):
    h5py = sys.modules['h5py']
    """Subclassed Model (ffn.py:237-258): model.layers = [sequential, concatenate, sequential_1, sequential_2]."""
    with h5py.File(path, "w") as f:
        f.attrs["layer_names"] = [b"sequential", b"concatenate", b"sequential_1", b"sequential_2"]
        f.create_group("concatenate").attrs["weight_names"] = []
        for grp, dense, bnname, kern, bn in (("sequential", "dense", "batch_normalization", w["w1"], w["bn1"]),
                                             ("sequential_1", "dense_1", "batch_normalization_1", w["w2"], w["bn2"])):
            g = f.create_group(grp)
            keys = (("gamma", "gamma"), ("beta", "beta"), ("moving_mean", "mean"), ("moving_variance", "var"))
            g.attrs["weight_names"] = [f"{dense}/kernel:0".encode()] + [f"{bnname}/{k}:0".encode() for k, _ in keys]
            g.create_dataset(f"{dense}/kernel:0", data=kern)
            for k, mine in keys:
                g.create_dataset(f"{bnname}/{k}:0", data=bn[mine])
        g = f.create_group("sequential_2")
        g.attrs["weight_names"] = [b"dense_2/kernel:0", b"dense_2/bias:0"]
        g.create_dataset("dense_2/kernel:0", data=w["w3"]); g.create_dataset("dense_2/bias:0", data=w["b3"])


@pytest.mark.parametrize("name", ("unet3_a", "unet3_b", "unet3_c"))
@pytest.mark.parametrize("full_model", (False, True))
def test_unet_round_trip(tmp_path, name, full_model):
    w = synth.make_unet_weights(name, seed=3)
    p = tmp_path / "unet.h5"
    _write_unet(p, w, full_model)
    got = kh5.read_unet_h5(p, ARCHS[name])
    for a, b in zip(got["convs"], w["convs"]):
        for k in ("kernel", "bias", "gamma", "beta", "mean", "var"):
            assert np.array_equal(a[k], b[k]), k
    assert np.array_equal(got["head"]["kernel"], w["head"]["kernel"]) and np.array_equal(got["head"]["bias"], w["head"]["bias"])
    wrong = "unet3_c" if name == "unet3_b" else "unet3_b"
    with pytest.raises(ValueError):
        kh5.read_unet_h5(p, ARCHS[wrong])


def test_unet_misordered_batchnorm_raises(tmp_path):
    w = synth.make_unet_weights("unet3_a", seed=3)
    w["convs"][2], w["convs"][3] = dict(w["convs"][2]), dict(w["convs"][3])
    w["convs"][3]["gamma"] = w["convs"][3]["gamma"][:16]                     # a BatchNormalization of the wrong width after conv 3
    p = tmp_path / "bad.h5"
    _write_unet(p, w, False)
    with pytest.raises(ValueError, match="wide"):
        kh5.read_unet_h5(p, ARCHS["unet3_a"])


def test_ffn_round_trip(tmp_path):
    w = synth.make_ffn_weights(seed=5)
    p = tmp_path / "ffn.h5"
    _write_ffn(p, w)
    got = kh5.read_ffn_h5(p)
    for k in ("w1", "w2", "w3", "b3"):
        assert np.array_equal(got[k], w[k])
    for b in ("bn1", "bn2"):
        for k in ("gamma", "beta", "mean", "var"):
            assert np.array_equal(got[b][k], w[b][k])
    with pytest.raises(ValueError):
        kh5.read_unet_h5(p, ARCHS["unet3_a"])


def test_real_h5py_cases_run_under_the_second_interpreter():
    """h5py is not importable by the image's main interpreter (the real-h5py cases above are skipped there), but /opt/conda/bin/python3.9
    has h5py 3.3: run this very module under it, so that every case executes against real HDF5 files wherever that interpreter exists."""
    import os
    import subprocess
    from pathlib import Path
    py = "/opt/conda/bin/python3.9"
    if not os.path.exists(py) or "CT_H5_INNER" in os.environ:
        pytest.skip("no second interpreter with h5py on this machine" if "CT_H5_INNER" not in os.environ else "inner run")
    import importlib.machinery
    if importlib.machinery.PathFinder.find_spec("h5py") is not None:          # (sys.modules may hold this module's stand-in: ask the path finder)
        pytest.skip("h5py is importable here: the cases above already ran against it")
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1", CT_H5_INNER="1")
    r = subprocess.run([py, "-W", "ignore", "-m", "pytest", str(Path(__file__)), "-q", "-p", "no:cacheprovider"], capture_output=True, text=True,
                       timeout=300, cwd=str(Path(__file__).resolve().parent.parent), env=env)
    assert r.returncode == 0 and " passed" in r.stdout and "skipped" in r.stdout.splitlines()[-1], r.stdout[-800:] + r.stderr[-400:]
    last = r.stdout.strip().splitlines()[-1]
    assert last.startswith("16 passed"), last          # (all round-trip / validation cases with both back ends; this wrapper skips itself inside)
