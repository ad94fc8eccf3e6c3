"""Host-side mirror of the reference's ``CellTracker/ffn.py`` inference surface.

    FFN()                      reference ffn.py:225-265 (callable, .predict, .load_weights)
    initial_matching_ffn(...)  reference ffn.py:268-327
    normalize_points(...)      reference ffn.py:330-374

The kNN shape features, the two dense layers and the all-pairs score kernel run in
csrc/ct_match.hip.  The (m*n) x 122 pair grid the reference tiles on the host is never built
(it is only materialised when a *foreign* model object exposing `.predict` is passed in).
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path
from typing import Tuple, Union

import numpy as np

from . import _dev, _lib
from .arch import FFN_FEAT, FFN_K_PTRS

RATIO_SEG_ERROR = 0.15
FFN_WEIGHTS_NAME = "weights_training_"
k_ptrs = FFN_K_PTRS
NUMBER_FEATURES = FFN_FEAT


def flatten_ffn_weights(w: dict) -> np.ndarray:
    parts = [w["w1"]] + [w["bn1"][k] for k in ("gamma", "beta", "mean", "var")] + [w["w2"]] + \
            [w["bn2"][k] for k in ("gamma", "beta", "mean", "var")] + [w["w3"], w["b3"]]
    return np.ascontiguousarray(np.concatenate([np.asarray(p, dtype=np.float32).ravel() for p in parts]))


class FFN:
    """Dense(61->512, no bias)+BN+LeakyReLU shared by both halves, concat, Dense(1024->512, no
    bias)+BN+LeakyReLU, Dense(512->1)+sigmoid -- evaluated by the HIP library."""

    def __init__(self, device: int | None = None):
        self._handle = None
        self._device = device
        self._weights = None

    def set_weights_dict(self, w: dict):
        t = _lib.require_gpu()
        if self._device is None:
            self._device = t.cuda.current_device()
        flat = flatten_ffn_weights(w)
        L = _lib.lib()
        if flat.size != L.ct_ffn_num_weights():
            raise ValueError(f"FFN weight count {flat.size} != {L.ct_ffn_num_weights()}")
        self._free()
        h = C.c_void_p()
        _lib.check(L.ct_ffn_create(flat.ctypes.data, flat.size, self._device, C.byref(h)), "ct_ffn_create")
        self._handle = h
        self._weights = w
        return self

    def load_weights(self, path):
        path = Path(path)
        if not path.exists():
            raise OSError(f"Unable to open file {path}")
        if path.suffix == ".npz":
            z = np.load(path)
            w = {"w1": z["w1"], "w2": z["w2"], "w3": z["w3"], "b3": z["b3"],
                 "bn1": {k: z[f"bn1_{k}"] for k in ("gamma", "beta", "mean", "var")},
                 "bn2": {k: z[f"bn2_{k}"] for k in ("gamma", "beta", "mean", "var")}}
            return self.set_weights_dict(w)
        try:
            import h5py  # noqa: F401
        except ImportError as e:
            raise OSError(f"cannot read {path}: Keras .h5 import needs h5py, which is not installed") from e
        from .keras_h5 import read_ffn_h5
        return self.set_weights_dict(read_ffn_h5(path))

    def save_weights(self, path):
        w = self._weights
        if w is None:
            raise ValueError("model has no weights")
        out = {"w1": w["w1"], "w2": w["w2"], "w3": w["w3"], "b3": w["b3"]}
        for b in ("bn1", "bn2"):
            for k in ("gamma", "beta", "mean", "var"):
                out[f"{b}_{k}"] = w[b][k]
        np.savez(path, **out)

    def _require(self):
        if self._handle is None:
            raise ValueError("FFN has no weights: call load_weights() or set_weights_dict() first")

    def pairgrid_device(self, feat_ref_d, feat_tgt_d):
        """feat_* fp32 device [n][61], [m][61] -> corr fp32 device [m][n] (async)."""
        t = _lib.require_gpu(); L = _lib.lib()
        self._require()
        n, m = feat_ref_d.shape[0], feat_tgt_d.shape[0]
        corr = _dev.empty((m, n), t.float32, feat_ref_d.device)
        ws = _dev.workspace(L.ct_ffn_workspace_bytes(n, m), feat_ref_d.device)
        _lib.check(L.ct_ffn_pairgrid(self._handle, feat_ref_d.data_ptr(), n, feat_tgt_d.data_ptr(), m, corr.data_ptr(),
                                     ws.data_ptr(), ws.numel(), _dev.stream(feat_ref_d.device)), "ct_ffn_pairgrid")
        return corr

    def predict(self, x, batch_size=None, verbose=0, **_):
        """x: (P,122) array, or the legacy two-input list [ (P,61), (P,61) ] (track.py:175) -> (P,1) float32."""
        t = _lib.require_gpu(); L = _lib.lib()
        self._require()
        if isinstance(x, (list, tuple)):
            x = np.concatenate([np.asarray(x[0]), np.asarray(x[1])], axis=1)
        x = np.asarray(x)
        if x.ndim != 2 or x.shape[1] != 2 * FFN_FEAT:
            raise ValueError(f"expected input of shape (P, {2 * FFN_FEAT}), got {x.shape}")
        out = np.empty((x.shape[0], 1), dtype=np.float32)
        step = 1 << 18
        for s in range(0, x.shape[0], step):
            xd = _dev.to_dev(x[s:s + step], t.float32, f"cuda:{self._device}")
            rows = xd.shape[0]
            od = _dev.empty((rows,), t.float32, xd.device)
            ws = _dev.workspace(L.ct_ffn_predict_workspace_bytes(rows), xd.device)
            _lib.check(L.ct_ffn_predict(self._handle, xd.data_ptr(), rows, od.data_ptr(), ws.data_ptr(), ws.numel(),
                                        _dev.stream(xd.device)), "ct_ffn_predict")
            out[s:s + rows, 0] = od.cpu().numpy()
        return out

    __call__ = predict

    def _free(self):
        if self._handle is not None:
            _lib.lib().ct_ffn_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self._free()
        except Exception:
            pass


def initial_matching_device(ffn_model: FFN, ref_d, tgt_d, k: int = FFN_K_PTRS):
    """ref_d, tgt_d: fp64 device [n][3], [m][3] -> corr fp32 device [m][n]."""
    return ffn_model.pairgrid_device(_dev.knn_features(ref_d, k), _dev.knn_features(tgt_d, k))


def initial_matching_ffn(ffn_model, ref: np.ndarray, tgt: np.ndarray, k_ptrs: int, two_inputs: bool = False) -> np.ndarray:
    """reference ffn.py:268-327 -> (m, n) float32, corr[t, r]."""
    ref_d, tgt_d = _dev.points_dev(ref), _dev.points_dev(tgt)
    if isinstance(ffn_model, FFN):
        return initial_matching_device(ffn_model, ref_d, tgt_d, k_ptrs).cpu().numpy()
    # foreign model object: features on the GPU, pair grid laid out exactly like the reference's
    fr = _dev.knn_features(ref_d, k_ptrs).cpu().numpy()
    ft = _dev.knn_features(tgt_d, k_ptrs).cpu().numpy()
    n, m = fr.shape[0], ft.shape[0]
    left = np.broadcast_to(fr[None], (m, n, fr.shape[1])).reshape(m * n, -1)
    right = np.broadcast_to(ft[:, None], (m, n, ft.shape[1])).reshape(m * n, -1)
    grid = [left, right] if two_inputs else np.concatenate([left, right], axis=1)
    return np.reshape(ffn_model.predict(grid, batch_size=1024), (m, n))


def normalize_points(points: np.ndarray, return_para: bool = False) -> Union[np.ndarray, Tuple[np.ndarray, Tuple]]:
    """reference ffn.py:330-374: centre, divide by 3 x std (ddof 0) of the projection on the first principal axis
    (device kernel: mean, 3x3 scatter matrix, largest eigenvalue by Jacobi rotations)."""
    points = np.asarray(points)
    if points.ndim != 2:
        raise ValueError(f"Points should be a 2D table, but get {points.ndim}D")
    if points.shape[1] != 3:
        raise ValueError(f"Points should have 3D coordinates, but get {points.shape[1]}D")
    out_d, para_d = _dev.normalize_points(_dev.points_dev(points))
    para = para_d.cpu().numpy()
    norm_points = out_d.cpu().numpy()
    return (norm_points, (para[:3].copy(), float(para[3]))) if return_para else norm_points
