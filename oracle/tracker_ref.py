"""ORACLE (test infrastructure -- never imported by the product path).

numpy restatement of the per-frame chain of the reference's legacy ``Tracker`` (CellTracker/tracker.py):

* get_subregions            <- CellTracker/track.py:501-533 (+ _get_coordinates :536-572), as used by cal_subregions :1095-1112
* transform_cells_quick     <- tracker.py:1350-1389
* correction_once_interp    <- tracker.py:1310-1348
* evaluate_correction       <- tracker.py:1402-1413
* accurate_correction       <- tracker.py:1177-1191
* get_cells_on_boundary     <- tracker.py:1291-1308
* segment_from_prob         <- the part of _segment (:636-650) after the U-Net, with the region step the product implements
                               (threshold + connected components, oracle/segment_ref.py) in place of the skimage watershed
* match_frame               <- tracker.py:1138-1175 (match) with _segment's U-Net half supplied by the caller

PARITY STATUS: PINNED -- tests/golden/make_golden.py runs the reference's own Tracker._accurate_correction /
_correction_once_interp / _get_cells_onBoundary / match (with _segment and the drawing replaced, as documented there) on
synthetic state and tests/test_oracle_tracker.py holds these functions to the recorded outputs (integers exact, fp64 1e-10).
The watershed region step of the reference is NOT restated (skimage is absent: parity unpinned).
"""
from __future__ import annotations

import numpy as np

from . import match_ref as mr
from . import segment_ref as sr

REP_NUM_PRGLS = 5
REP_NUM_CORRECTION = 20
BOUNDARY_XY = 6


def get_subregions(label_image: np.ndarray, num: int):
    region_list, region_width, region_min = [], [], []
    for lab in range(1, num + 1):
        idx = np.where(label_image == lab)
        lo = [int(np.min(a)) for a in idx]; hi = [int(np.max(a)) for a in idx]
        region_list.append(label_image[lo[0]:hi[0] + 1, lo[1]:hi[1] + 1, lo[2]:hi[2] + 1] == lab)
        region_width.append([hi[d] + 1 - lo[d] for d in range(3)])
        region_min.append(lo)
    return region_list, region_width, region_min


class LegacyState:
    """What cal_subregions / interpolate_seg leave on the Tracker instance (tracker.py:1046-1112)."""

    def __init__(self, seg_interp: np.ndarray, siz_xyz, z_xy_ratio: float, z_scaling: int, tracked_t0=None):
        self.x_siz, self.y_siz, self.z_siz = (int(v) for v in siz_xyz)
        self.z_xy_ratio = float(z_xy_ratio)
        self.z_scaling = int(z_scaling)
        self.seg_interp = np.asarray(seg_interp)
        seg16 = self.seg_interp.astype("int16")
        self.n_cells = int(seg16.max())
        self.region_list, self.region_width, self.region_xyz_min = get_subregions(seg16, self.n_cells)
        self.pad = tuple(int(v) for v in np.max(self.region_width, axis=0))
        if tracked_t0 is None:
            from scipy import ndimage
            z_range = range(self.z_scaling // 2, self.seg_interp.shape[2], self.z_scaling)
            layer = self.seg_interp[:, :, z_range]
            cen = np.asarray(ndimage.center_of_mass(layer > 0, layer, range(1, int(layer.max()) + 1)))
            tracked_t0 = cen * np.array([1.0, 1.0, self.z_xy_ratio])
        self.tracked_t0 = np.asarray(tracked_t0, dtype=np.float64)


def transform_cells_quick(st: LegacyState, vectors3d: np.ndarray):
    px, py, pz = st.pad
    shape = tuple(s + 2 * p for s, p in zip(st.seg_interp.shape, st.pad))
    label_moved = np.zeros(shape, dtype=np.int16)
    mask = np.zeros(shape, dtype=np.int16)
    for lab in range(len(st.region_list)):
        lo = [st.region_xyz_min[lab][d] + int(vectors3d[lab, d]) + st.pad[d] for d in range(3)]
        w = st.region_width[lab]
        sl = tuple(slice(lo[d], lo[d] + w[d]) for d in range(3))
        prev = label_moved[sl]
        if prev.shape != st.region_list[lab].shape:
            continue
        label_moved[sl] = prev * (1 - st.region_list[lab]) + st.region_list[lab] * (lab + 1)
        mask[sl] += (st.region_list[lab] > 0).astype("int8")
    return label_moved[px:-px, py:-py, pz:-pz], mask[px:-px, py:-py, pz:-pz]


def _center_of_mass(weight: np.ndarray, labels: np.ndarray, n: int) -> np.ndarray:
    """scipy.ndimage.center_of_mass(weight, labels, range(1, n + 1)) -> (n, 3), NaN rows for absent labels."""
    flat = labels.ravel().astype(np.int64)
    w = weight.ravel().astype(np.float64)
    norm = np.bincount(flat, weights=w, minlength=n + 1)[1:n + 1]
    out = np.empty((n, 3))
    grids = np.ogrid[[slice(0, s) for s in labels.shape]]
    with np.errstate(invalid="ignore", divide="ignore"):
        for d in range(3):
            g = np.broadcast_to(grids[d].astype(float), labels.shape).ravel()
            out[:, d] = np.bincount(flat, weights=w * g, minlength=n + 1)[1:n + 1] / norm
    return out


def correction_once_interp(st: LegacyState, image_cell_bg_xyz, image_gcn, i_disp: np.ndarray, cell_on_bound: np.ndarray):
    lab_i, ovl_i = transform_cells_quick(st, i_disp)
    zs = slice(st.z_scaling // 2, st.z_siz * st.z_scaling, st.z_scaling)
    labels = lab_i[:, :, zs].copy(); overlap = ovl_i[:, :, zs]
    labels[overlap > 1] = 0
    for i in np.where(cell_on_bound == 1)[0]:
        labels[labels == (i + 1)] = 0
    l_move = st.tracked_t0 * np.array([1, 1, 1 / st.z_xy_ratio]) + i_disp * np.array([1, 1, 1 / st.z_scaling])
    com = _center_of_mass(image_cell_bg_xyz + image_gcn, labels, st.n_cells)
    lost = np.isnan(com[:, 0])
    corr = com - l_move
    corr[lost, :] = 0
    corr[:, 2] = corr[:, 2] * st.z_xy_ratio
    r_disp = i_disp * np.array([1, 1, st.z_xy_ratio / st.z_scaling]) + corr
    i_new = real_to_interpolated(st, r_disp)
    return r_disp, i_new, corr


def real_to_interpolated(st: LegacyState, r_disp):
    d = np.array(r_disp).copy()
    d[:, 2] = d[:, 2] * (st.z_scaling / st.z_xy_ratio)
    return np.rint(d).astype(int)


def evaluate_correction(st: LegacyState, corr: np.ndarray) -> bool:
    t = corr.copy()
    t[:, 2] *= st.z_scaling / st.z_xy_ratio
    return not (np.nanmax(np.abs(t)) >= 0.5)


def accurate_correction(st: LegacyState, image_cell_bg_xyz, image_gcn, r_disp_prev, r_tracked_prev, cells_on_boundary, r_coor_predicted,
                        return_rounds=False):
    r_disp = r_disp_prev + (r_coor_predicted - r_tracked_prev)
    i_disp = real_to_interpolated(st, r_disp)
    rounds = 0
    for i in range(REP_NUM_CORRECTION):
        r_disp, i_disp, corr = correction_once_interp(st, image_cell_bg_xyz, image_gcn, i_disp, cells_on_boundary)
        rounds += 1
        if i == REP_NUM_CORRECTION - 1 or evaluate_correction(st, corr):
            break
    return (r_disp, i_disp, rounds) if return_rounds else (r_disp, i_disp)


def get_cells_on_boundary(st: LegacyState, r_coords: np.ndarray, ensemble) -> np.ndarray:
    b = 0 if ensemble else BOUNDARY_XY
    z = r_coords[:, 2] / st.z_xy_ratio
    bad = (r_coords[:, 0] < b) | (r_coords[:, 1] < b) | (r_coords[:, 0] > st.x_siz - b) | (r_coords[:, 1] > st.y_siz - b) | (z < 0) | (z > st.z_siz)
    return np.where(bad)[0]


def segment_from_prob(image_cell_bg_xyz: np.ndarray, z_xy_ratio: float, min_size: int, connectivity: int = 1, region_method: str = "watershed"):
    """-> (l_center_coordinates (n, 3), segmentation_auto int32, r_coordinates_segment): the tail of _segment (:640-648).  Region step:
    the reference's marker watershed (oracle/watershed_ref.py, method "min_size") or, region_method "cc", connected components."""
    if region_method == "watershed":
        from oracle import watershed_ref as wr
        labels, centres, _, _ = wr.segment_centroids(np.asarray(image_cell_bg_xyz, dtype=np.float32), z_xy_ratio, "min_size", min_size)
    else:
        labels, centres, _ = sr.segment_centroids(image_cell_bg_xyz, 0.5, connectivity, min_size)
    r = np.array(centres).copy()
    r[:, 2] = r[:, 2] * z_xy_ratio
    return centres, labels, r


def match_frame(st: LegacyState, ffn_predict, image_cell_bg_xyz, image_raw, r_seg_t0, min_size, beta_tk, lambda_tk, maxiter_tk,
                ensemble=False, cells_on_boundary=None):
    """Tracker.match (:1138-1175) for volume 1 -> target, image_cell_bg supplied by the caller.
    -> dict(r_coordinates_segment, r_coor_predicted, cells_on_boundary_local, r_disp, i_disp)."""
    image_gcn = image_raw.copy() / 65536.0
    _, seg_auto, r_seg = segment_from_prob(image_cell_bg_xyz, st.z_xy_ratio, min_size)
    pred = mr.predict_pos_once(ffn_predict, r_seg_t0, st.tracked_t0, r_seg, beta_tk, lambda_tk, maxiter_tk)
    bd = get_cells_on_boundary(st, pred, ensemble)
    local = (np.zeros(st.n_cells, dtype=int) if cells_on_boundary is None else cells_on_boundary.copy())
    local[bd] = 1
    r_disp, i_disp = accurate_correction(st, image_cell_bg_xyz, image_gcn, np.zeros((st.n_cells, 3)), st.tracked_t0, local, pred)
    return {"r_coordinates_segment": r_seg, "segmentation_auto": seg_auto, "r_coor_predicted": pred,
            "cells_on_boundary_local": local, "r_disp": r_disp, "i_disp": i_disp}
