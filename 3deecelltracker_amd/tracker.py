"""Matching half of the reference's legacy ``CellTracker/tracker.py`` `Tracker`.

Accelerated here (reference tracker.py):
    match                 :1138-1175
    _predict_cellregions / _save_unet_regions :652-669   (LCN -> U-Net -> unet_cache/t%06i.npy float16)
    segment_prob          the part of _segment (:636-650) after the U-Net, with connected components instead of the
                          skimage watershed (segment.py)
    _predict_pos_once     :1193-1222   draw=False branch
    _fit_ffn_prgls        :1224-1254
    _ffn_prgls_once       :1256-1267
    _predict_one_rep      :1269-1289
    _get_cells_onBoundary :1291-1308
    track_one_vol (ensemble part) :1499-1509

REP_NUM_PRGLS x (FFN -> legacy PR-GLS with beta * 0.8^i) chained on the device, then the fields
are re-applied to the tracked coordinates; nothing returns to the host in between.
"""
from __future__ import annotations

import os
from functools import reduce
from types import SimpleNamespace

import numpy as np

from . import _dev
from .ffn import FFN, initial_matching_device
from .track import get_reference_vols, initial_matching_quick, pr_gls_quick

REP_NUM_PRGLS = 5
REP_NUM_CORRECTION = 20
BOUNDARY_XY = 6


class Tracker:
    ensemble_chains = 4      # source-volume predictions of one ensemble step in flight on this GPU (parallel.chain_map)

    def __init__(self, ffn_model, beta_tk=300, lambda_tk=0.1, max_iteration=20, ensemble=False, adjacent=False,
                 volume_shape=None, z_xy_ratio=1.0, miss_frame=None, unet_model=None, noise_level=None, shrink=(24, 24, 2),
                 unet_cache=None):
        self.ffn_model = ffn_model
        # segmentation half (optional): the U-Net, its pre-processing and the reference's on-disk cache of its output
        self.unet_model = unet_model
        self.noise_level = noise_level
        self.shrink = tuple(shrink)
        self.paths = SimpleNamespace(unet_cache=None if unet_cache is None else os.path.join(str(unet_cache), ""))
        self.beta_tk = beta_tk
        self.lambda_tk = lambda_tk
        self.max_iteration = max_iteration
        self.ensemble = ensemble
        self.adjacent = adjacent
        self.miss_frame = [] if not miss_frame else miss_frame
        self.z_xy_ratio = z_xy_ratio
        if volume_shape is not None:
            self.x_siz, self.y_siz, self.z_siz = volume_shape
        self.history = SimpleNamespace(r_segmented_coordinates=[], r_tracked_coordinates=[])
        self.segresult = SimpleNamespace(r_coordinates_segment=None)
        self.cell_num_t0 = 0

    # ---- state the segmentation half would fill in
    def set_volume1(self, r_segmented_coordinates, r_tracked_coordinates=None):
        seg = np.asarray(r_segmented_coordinates, dtype=np.float64)
        trk = seg.copy() if r_tracked_coordinates is None else np.asarray(r_tracked_coordinates, dtype=np.float64)
        self.history.r_segmented_coordinates = [seg]
        self.history.r_tracked_coordinates = [trk]
        self.cell_num_t0 = trk.shape[0]

    def set_segmentation(self, r_coordinates_segment):
        self.segresult.r_coordinates_segment = np.asarray(r_coordinates_segment, dtype=np.float64)

    def _predict_cellregions(self, image_raw, vol):
        """reference :652-660: the U-Net output of volume `vol`, from unet_cache/t%06i.npy when it is there."""
        if self.paths.unet_cache is not None:
            try:
                return np.load(self.paths.unet_cache + "t%06i.npy" % vol, allow_pickle=True)
            except OSError:
                pass
        return self._save_unet_regions(image_raw, vol)

    def _save_unet_regions(self, image_raw, vol):
        """reference :662-669: _normalize_image -> unet3_prediction, cached as float16 [1, x, y, z, 1]."""
        from .preprocess import _normalize_image
        from .unet3d import unet3_prediction
        if self.unet_model is None or self.noise_level is None:
            raise ValueError("Tracker was created without unet_model / noise_level: no segmentation half")
        image_norm = np.expand_dims(_normalize_image(image_raw, self.noise_level), axis=(0, 4))
        image_cell_bg = unet3_prediction(image_norm, self.unet_model, shrink=self.shrink)
        if self.paths.unet_cache is not None:
            os.makedirs(self.paths.unet_cache, exist_ok=True)
            np.save(self.paths.unet_cache + "t%06i.npy" % vol, np.array(image_cell_bg, dtype="float16"))
        return image_cell_bg

    def segment_prob(self, image_cell_bg, min_size=0, threshold=0.5, connectivity=1):
        """The part of reference :636-650 (_segment) after the U-Net: regions -> centres -> real coordinates, on the GPU.

        Regions come from threshold + connected components (segment.py; the skimage watershed is not rebuilt);
        centres = center_of_mass(regions > 0, regions, 1..n) (:646), r = _transform_layer_to_real (:559-561, z * z_xy_ratio).
        Returns (l_center_coordinates, segmentation_auto, r_coordinates_segment) and records the segmentation for match()."""
        from .segment import segment_centroids
        labels, l_centres, _ = segment_centroids(image_cell_bg, threshold, connectivity, min_size)
        r = l_centres.copy()
        r[:, 2] *= self.z_xy_ratio
        self.set_segmentation(r)
        return l_centres, labels, r

    # ---- matching
    def match(self, target_volume, r_coordinates_segment=None, method="min_size"):
        """reference :1138-1175.  Returns (None, [cells_on_boundary, target_volume, None, r_coor_predicted])."""
        if target_volume in self.miss_frame:
            raise ValueError("target_volume is a miss_frame")
        if r_coordinates_segment is not None:
            self.set_segmentation(r_coordinates_segment)
        if self.segresult.r_coordinates_segment is None:
            raise ValueError("no segmentation for the target volume: pass r_coordinates_segment")
        r_coor_predicted, anim = self._predict_pos_once(source_volume=1, draw=False)
        cells_on_boundary = np.zeros(self.cell_num_t0, dtype=int)
        if hasattr(self, "x_siz"):
            cells_on_boundary[self._get_cells_onBoundary(r_coor_predicted, self.ensemble)] = 1
        return anim, [cells_on_boundary, target_volume, None, r_coor_predicted]

    def _fit_device(self, seg_pre_d, seg_tgt_d, rep):
        C_t, beta_t, inter_t = [], [], []
        inter = seg_pre_d
        for i in range(rep):
            beta = self.beta_tk * (0.8 ** i)
            inter_t.append(inter)
            if isinstance(self.ffn_model, FFN):
                corr = initial_matching_device(self.ffn_model, inter, seg_tgt_d, 20)
            else:
                corr = _dev.to_dev(initial_matching_quick(self.ffn_model, inter.cpu().numpy(), seg_tgt_d.cpu().numpy(), 20),
                                   _dev.torch().float32)
            _, moved, C = _dev.prgls_legacy(inter, seg_tgt_d, corr, beta, self.max_iteration, self.lambda_tk, 1e8, want_P=False)
            inter = moved
            C_t.append(C); beta_t.append(beta)
        return C_t, beta_t, inter_t

    def _predict_pos_once(self, source_volume, draw=False):
        """reference :1193-1222"""
        if draw:
            raise NotImplementedError("animation drawing is outside the accelerated path")
        seg_pre = _dev.points_dev(self.history.r_segmented_coordinates[source_volume - 1])
        seg_tgt = _dev.points_dev(self.segresult.r_coordinates_segment)
        C_t, beta_t, inter_t = self._fit_device(seg_pre, seg_tgt, REP_NUM_PRGLS)
        pred = _dev.points_dev(self.history.r_tracked_coordinates[source_volume - 1]).clone()
        for C, b, inter in zip(C_t, beta_t, inter_t):
            _dev.gram_apply(pred, inter, C, b)
        return pred.cpu().numpy(), None

    def _fit_ffn_prgls(self, rep, r_coordinates_segment_pre):
        """reference :1224-1254 -> (C_t, BETA_t, coor_intermediate_list) as numpy."""
        C_t, beta_t, inter_t = self._fit_device(_dev.points_dev(r_coordinates_segment_pre),
                                                _dev.points_dev(self.segresult.r_coordinates_segment), rep)
        return [c.cpu().numpy() for c in C_t], beta_t, [x.cpu().numpy() for x in inter_t]

    def _ffn_prgls_once(self, i, r_coordinates_segment_pre):
        """reference :1256-1267"""
        init_match = initial_matching_quick(self.ffn_model, r_coordinates_segment_pre,
                                            self.segresult.r_coordinates_segment, 20)
        P, post, C = pr_gls_quick(np.array(r_coordinates_segment_pre, copy=True), self.segresult.r_coordinates_segment,
                                  init_match, BETA=self.beta_tk * (0.8 ** i), max_iteration=self.max_iteration,
                                  LAMBDA=self.lambda_tk)
        return C, post

    def _predict_one_rep(self, r_coordinates_predicted_pre, coor_intermediate_list, BETA_t, C_t):
        """reference :1269-1289"""
        t = _dev.torch()
        pred = _dev.points_dev(r_coordinates_predicted_pre).clone()
        _dev.gram_apply(pred, _dev.points_dev(coor_intermediate_list), _dev.to_dev(np.asarray(C_t, dtype=np.float64), t.float64), BETA_t)
        return pred.cpu().numpy(), r_coordinates_predicted_pre

    def _get_cells_onBoundary(self, r_coordinates_prgls, ensemble):
        """reference :1291-1308"""
        boundary_xy = 0 if ensemble else BOUNDARY_XY
        return np.where(reduce(np.logical_or, [
            r_coordinates_prgls[:, 0] < boundary_xy, r_coordinates_prgls[:, 1] < boundary_xy,
            r_coordinates_prgls[:, 0] > self.x_siz - boundary_xy, r_coordinates_prgls[:, 1] > self.y_siz - boundary_xy,
            r_coordinates_prgls[:, 2] / self.z_xy_ratio < 0, r_coordinates_prgls[:, 2] / self.z_xy_ratio > self.z_siz]))

    def predict_ensemble(self, vol, source_vols=None):
        """Ensemble part of track_one_vol (reference :1499-1509): every source volume's prediction of
        the *displacement-corrected* positions, trim-mean'd.  `history` must hold all source volumes."""
        from . import parallel
        t = _dev.torch()
        vols = get_reference_vols(self.ensemble, vol, adjacent=self.adjacent) if source_vols is None else source_vols

        def one(v):
            pred, _ = self._predict_pos_once(source_volume=v, draw=False)
            return _dev.to_dev(pred, t.float64)
        stack = parallel.sharded_map_gather(one, vols, chains=self.ensemble_chains)
        return _dev.trim_mean(stack, 0.1).cpu().numpy()
