"""Host-side mirror of the reference's legacy ``CellTracker/tracker.py`` `Tracker` (same constructor, same method names).

Per-frame chain, all on the device and on one stream (reference tracker.py):

    match / track_one_vol          :1138-1175, :1473-1536
      _segment                     :605-650    raw stack -> image_gcn, LCN -> U-Net (or unet_cache/t%06i.npy) -> regions -> centres
        _predict_cellregions / _save_unet_regions :652-669   (ct_normalize_image, ct_unet_predict_volume; float16 cache file)
        _watershed                 :671-684    ct_watershed_segment: watershed_2d + watershed_3d + relabel_sequential on the device
                                               (pinned against the reference on scikit-image 0.18.3: tests/test_watershed_pin.py);
                                               region_method = "cc" selects threshold + connected components instead
      _predict_pos_once            :1193-1222  REP_NUM_PRGLS x (FFN -> legacy PR-GLS with beta * 0.8^i), fields re-applied
      _get_cells_onBoundary        :1291-1308
      _accurate_correction         :1177-1191  ct_accurate_correction_legacy (_correction_once_interp :1310-1348,
                                               _transform_cells_quick :1350-1389, _evaluate_correction :1402-1413)
    ensemble part of track_one_vol :1499-1509  source volumes sharded over ranks / run as concurrent chains, device trim_mean

Host logic kept like the reference: Paths / folders (:687-753), SegResults, History, set_segmentation, set_tracking,
initiate_tracking, cal_subregions (track.get_subregions), _reset_tracking_state, save_coordinates, the unit transforms.
Image files are read with PIL (the reference uses tifffile); `image_reader` / `inject_*` are injection points for callers
that hold their stacks in memory.  Not accelerated, raise NotImplementedError: drawing (`Draw`), U-Net re-training,
`interpolate_seg` when smoothed cells overlap (needs skimage's watershed) and the tracked *label images*
(`_transform_motion_to_image` goes through the same watershed); coordinates are complete without them.
"""
from __future__ import annotations

import ctypes as C
import os
from functools import reduce

import numpy as np

from . import _dev, _lib
from .ffn import FFN, initial_matching_device
from .track import get_reference_vols, initial_matching_quick, pr_gls_quick

REP_NUM_PRGLS = 5
REP_NUM_CORRECTION = 20
BOUNDARY_XY = 6


def get_tracking_path(adjacent, ensemble, folder_path):
    """reference :90-110"""
    if not ensemble:
        return os.path.join(folder_path, "track_results_SingleMode/")
    if not adjacent:
        return os.path.join(folder_path, "track_results_EnsembleDstrbtMode/")
    return os.path.join(folder_path, "track_results_EnsembleAdjctMode/")


def _make_folder(path_i):
    os.makedirs(path_i, exist_ok=True)
    return path_i


def read_image_ts(vol, path, name, z_range, print_=False):
    """reference :113-142 (tifffile.imread there; PIL here): (row, column, layer) array of volume `vol`."""
    from PIL import Image
    layers = [np.array(Image.open(path + name % (vol, z))) for z in range(z_range[0], z_range[1])]
    img_array = np.array(layers).transpose((1, 2, 0))
    if print_:
        print("Load images with shape:", img_array.shape)
    return img_array


def save_automatic_segmentation(labels_xyz, folder_path, use_8_bit: bool):
    """reference :145-165"""
    from PIL import Image
    os.makedirs(os.path.join(folder_path, "auto_vol1"), exist_ok=True)
    dtype = np.uint8 if use_8_bit else np.uint16
    for z in range(1, labels_xyz.shape[2] + 1):
        Image.fromarray(labels_xyz[:, :, z - 1].astype(dtype)).save(os.path.join(folder_path, "auto_vol1", "auto_vol1_z%04i.tif" % z))


class SegResults:
    """reference :464-496.  `image_cell_bg` / `segmentation_auto` are fetched from the device on first access."""

    def __init__(self):
        self._image_cell_bg = None
        self.image_cell_bg_d = None          # fp32 cuda [x, y, z]
        self.raw_d = None                    # the raw stack on the device (image_gcn = raw / 65536 is never materialised there)
        self.l_center_coordinates = None
        self._segmentation_auto = None
        self.segmentation_auto_d = None
        self._image_gcn = None
        self.r_coordinates_segment = None

    @property
    def image_cell_bg(self):
        if self._image_cell_bg is None and self.image_cell_bg_d is not None:
            self._image_cell_bg = self.image_cell_bg_d.cpu().numpy()[None, :, :, :, None]
        return self._image_cell_bg

    @property
    def segmentation_auto(self):
        if self._segmentation_auto is None and self.segmentation_auto_d is not None:
            self._segmentation_auto = self.segmentation_auto_d.cpu().numpy()
        return self._segmentation_auto

    @property
    def image_gcn(self):
        if self._image_gcn is None and self.raw_d is not None:
            self._image_gcn = self.raw_d.cpu().numpy().astype(np.float64) / 65536.0
        return self._image_gcn

    def update_results(self, image_cell_bg, l_center_coordinates, segmentation_auto, image_gcn, r_coordinates_segment):
        t = _dev.torch()
        self._image_cell_bg = self._segmentation_auto = self._image_gcn = None
        if hasattr(image_cell_bg, "is_cuda"):
            self.image_cell_bg_d = image_cell_bg
        else:
            self._image_cell_bg = image_cell_bg
            a = np.asarray(image_cell_bg)
            self.image_cell_bg_d = t.from_numpy(np.ascontiguousarray(a[0, :, :, :, 0] if a.ndim == 5 else a, dtype=np.float32)).cuda()
        if hasattr(segmentation_auto, "is_cuda"):
            self.segmentation_auto_d = segmentation_auto
        else:
            self._segmentation_auto = segmentation_auto; self.segmentation_auto_d = None
        if hasattr(image_gcn, "is_cuda"):
            self.raw_d = image_gcn                                   # device callers hand over the raw stack itself
        else:
            self._image_gcn = image_gcn
            self.raw_d = None if image_gcn is None else t.from_numpy(np.ascontiguousarray(np.asarray(image_gcn) * 65536.0, dtype=np.float32)).cuda()
        self.l_center_coordinates = l_center_coordinates
        self.r_coordinates_segment = r_coordinates_segment


class Paths:
    """reference :687-753"""

    def __init__(self, folder_path, image_name, unet_model_file, ffn_model_file):
        self.folder = folder_path
        self.models = self.unet_cache = self.raw_image = self.auto_segmentation_vol1 = None
        self.manual_segmentation_vol1 = self.unet_weights = self.track_results = self.track_information = self.anim = None
        self.image_name = image_name
        self.unet_model_file = unet_model_file
        self.ffn_model_file = ffn_model_file

    def make_folders(self, adjacent, ensemble):
        folder_path = self.folder
        self.raw_image = _make_folder(os.path.join(folder_path, "data/"))
        self.auto_segmentation_vol1 = _make_folder(os.path.join(folder_path, "auto_vol1/"))
        self.manual_segmentation_vol1 = _make_folder(os.path.join(folder_path, "manual_vol1/"))
        self.track_information = _make_folder(os.path.join(folder_path, "track_information/"))
        self.models = _make_folder(os.path.join(folder_path, "models/"))
        self.unet_cache = _make_folder(os.path.join(folder_path, "unet_cache/"))
        self.track_results = _make_folder(get_tracking_path(adjacent, ensemble, folder_path))
        self.anim = _make_folder(os.path.join(folder_path, "anim/"))
        self.unet_weights = _make_folder(os.path.join(self.models, "unet_weights/"))


class History:
    """reference :756-776"""

    def __init__(self):
        self.r_displacements = []
        self.r_segmented_coordinates = []
        self.r_tracked_coordinates = []
        self.anim = []


def _not_accelerated(name):
    def method(self, *a, **k):
        raise NotImplementedError(f"Tracker.{name}: drawing / re-training is outside the MI355X per-frame path")
    method.__name__ = name
    return method


class Tracker:
    """reference :779-1551.  Constructor and public method names follow the reference (:854-859)."""

    ensemble_chains = 8      # source-volume predictions of one ensemble step in flight on this GPU (parallel.chain_map)
    ensemble_batched = True  # ... or, when the point sets are small enough, all of them as one batched chain of launches
    region_method = "watershed"   # region step of _segment: the reference's marker watershed; "cc": threshold + connected components
    connectivity = 1         # region_method "cc": 6-connected components (scipy.ndimage.label default)

    def __init__(self, volume_num, siz_xyz: tuple, z_xy_ratio, z_scaling, noise_level, min_size, beta_tk, lambda_tk, maxiter_tk,
                 folder_path, image_name, unet_model_file, ffn_model_file, cell_num=0, ensemble=False, adjacent=False,
                 shrink=(24, 24, 2), miss_frame=None):
        self._init_state(volume_num, siz_xyz, z_xy_ratio, z_scaling, noise_level, min_size, beta_tk, lambda_tk, maxiter_tk,
                         cell_num, ensemble, adjacent, shrink, miss_frame)
        self.paths = Paths(folder_path, image_name, unet_model_file, ffn_model_file)
        self.paths.make_folders(adjacent, ensemble)

    def _init_state(self, volume_num, siz_xyz, z_xy_ratio, z_scaling, noise_level, min_size, beta_tk, lambda_tk, maxiter_tk,
                    cell_num, ensemble, adjacent, shrink, miss_frame):
        # Segmentation.__init__ (:504-518)
        self.volume_num = volume_num
        self.x_siz, self.y_siz, self.z_siz = (None, None, None) if siz_xyz is None else siz_xyz
        self.z_xy_ratio = z_xy_ratio
        self.z_scaling = z_scaling
        self.shrink = tuple(shrink)
        self.vol = None
        self.paths = None
        self.unet_model = None
        self.r_coordinates_segment_t0 = None
        self.segresult = SegResults()
        # Tracker.__init__ (:862-887)
        self.miss_frame = [] if not miss_frame else miss_frame
        self.noise_level = noise_level
        self.min_size = min_size
        self.beta_tk = beta_tk
        self.lambda_tk = lambda_tk
        self.max_iteration = maxiter_tk
        self.ensemble = ensemble
        self.adjacent = adjacent
        self.cell_num = cell_num
        self.cell_num_t0 = None
        self.Z_RANGE_INTERP = None
        self.region_list = self.region_width = self.region_xyz_min = None
        self.pad_x = self.pad_y = self.pad_z = None
        self.label_padding = None
        self.segmentation_manual_relabels = None
        self.seg_cells_interpolated_corrected = None
        self.r_coordinates_tracked_t0 = None
        self.cells_on_boundary = None
        self.ffn_model = None
        self.val_losses = None
        self.history = History()
        self.use_8_bit = True
        self.tracked_labels = None
        # this build
        self.image_reader = None             # callable(vol) -> (x, y, z) array; default: PIL on paths.raw_image + image_name
        self.last_correction_rounds = 0
        self._regions_dev = None
        self._injected = False               # inject_segmentation() supplies the next match's segmentation (one shot)

    @classmethod
    def for_matching(cls, ffn_model, beta_tk=300, lambda_tk=0.1, maxiter_tk=20, ensemble=False, adjacent=False, siz_xyz=None,
                     z_xy_ratio=1.0, z_scaling=1, miss_frame=None):
        """A Tracker without folders / U-Net, for callers that only use the matching half (_predict_pos_once, predict_ensemble)
        on coordinates they already hold."""
        self = object.__new__(cls)
        self._init_state(0, siz_xyz, z_xy_ratio, z_scaling, None, None, beta_tk, lambda_tk, maxiter_tk, 0, ensemble, adjacent,
                         (24, 24, 2), miss_frame)
        self.ffn_model = ffn_model
        return self

    # ------------------------------------------------------------------ parameters (reference :520-550, :889-906)
    def set_segmentation(self, noise_level=None, min_size=None, del_cache=False):
        if self.noise_level == noise_level and self.min_size == min_size:
            print("Segmentation parameters were not modified")
        elif noise_level is None and min_size is None:
            print("Segmentation parameters were not modified")
        else:
            if noise_level is not None:
                self.noise_level = noise_level
            if min_size is not None:
                self.min_size = min_size
            print(f"Parameters were modified: noise_level={self.noise_level}, min_size={self.min_size}")
            del_cache = True
        if del_cache:
            for f in os.listdir(self.paths.unet_cache):
                os.remove(os.path.join(self.paths.unet_cache, f))
            print("All files under /unet folder were deleted")

    def set_tracking(self, beta_tk, lambda_tk, maxiter_tk):
        if self.beta_tk == beta_tk and self.lambda_tk == lambda_tk and self.max_iteration == maxiter_tk:
            print("Tracking parameters were not modified")
        else:
            self.beta_tk, self.lambda_tk, self.max_iteration = beta_tk, lambda_tk, maxiter_tk
            print(f"Parameters were modified: beta_tk={self.beta_tk}, lambda_tk={self.lambda_tk}, maxiter_tk={self.max_iteration}")

    # ------------------------------------------------------------------ unit transforms (reference :552-573)
    @staticmethod
    def _transform_disps(disp, factor):
        """Coordinates / displacements with their z column multiplied by `factor` (x, y untouched); always a fresh array of the
        input's dtype: an integer array (the reference feeds _transform_real_to_interpolated's output back in, :307, :451) gets its
        scaled column truncated on assignment, as numpy does for the reference (:553-556) -- an in-place `*=` would raise instead."""
        scaled = np.array(disp, copy=True)
        scaled[:, 2] = scaled[:, 2] * factor
        return scaled

    def _z_units(self, disp, factor, to_voxels):
        out = self._transform_disps(disp, factor)
        return np.rint(out).astype(int) if to_voxels else out

    # the four unit changes of the reference (:557-573): "layer" = raw voxels, "real" = voxels with z in x/y units,
    # "interpolated" = voxels of the z-upsampled image; everything that lands on a voxel grid is rounded to int
    def _transform_layer_to_real(self, voxel_disp):
        return self._z_units(voxel_disp, self.z_xy_ratio, to_voxels=False)

    def _transform_real_to_interpolated(self, r_disp):
        return self._z_units(r_disp, self.z_scaling / self.z_xy_ratio, to_voxels=True)

    def _transform_real_to_layer(self, r_disp):
        return self._z_units(r_disp, 1 / self.z_xy_ratio, to_voxels=True)

    def _transform_interpolated_to_layer(self, r_disp):
        return self._z_units(r_disp, 1 / self.z_scaling, to_voxels=True)

    # ------------------------------------------------------------------ models (reference :575-581, :1119-1122)
    def load_unet(self):
        """models/<unet_model_file>: `.npz` (this package's save_weights) or a Keras `.h5` (needs h5py).  The reference also
        stores the initial weights for re-training (:580); they are written as .npz."""
        from .unet3d import load_model
        self.unet_model = load_model(os.path.join(self.paths.models, self.paths.unet_model_file))
        self.unet_model.save_weights(os.path.join(self.paths.unet_weights, "weights_initial.npz"))
        print("Loaded the 3D U-Net model")

    def load_ffn(self):
        path = os.path.join(self.paths.models, self.paths.ffn_model_file)
        self.ffn_model = FFN()
        self.ffn_model.load_weights(path)
        print("Loaded the FFN model")

    # ------------------------------------------------------------------ segmentation half
    def _read_image(self, vol, print_shape=False):
        if self.image_reader is not None:
            return np.asarray(self.image_reader(vol))
        return read_image_ts(vol, self.paths.raw_image, self.paths.image_name, (1, self.z_siz + 1), print_=print_shape)

    def segment_vol1(self, method="min_size"):
        """reference :583-603"""
        self.vol = 1
        self.segresult.update_results(*self._segment(self.vol, method=method, print_shape=True))
        self.r_coordinates_segment_t0 = self.segresult.r_coordinates_segment.copy()
        seg = self.segresult.segmentation_auto
        save_automatic_segmentation(labels_xyz=seg, folder_path=self.paths.folder, use_8_bit=bool(seg.max() <= 255))
        print("Segmented volume 1 and saved it")

    def _segment(self, vol, method, print_shape=False):
        """reference :605-650.  Returns (image_cell_bg, l_center_coordinates, segmentation_auto, image_gcn, r_coordinates_segment)
        where the three images are *device tensors* (prob fp32 [x, y, z], labels int32, the raw stack): SegResults converts
        them to the reference's numpy forms on access.  Nothing but the (n, 3) centres visits the host."""
        t = _dev.torch()
        image_raw = self._read_image(vol, print_shape)
        raw_d = t.from_numpy(np.ascontiguousarray(image_raw if image_raw.dtype == np.uint16 else image_raw.astype(np.float32))).cuda()
        prob_d = self._predict_cellregions_device(raw_d, vol)
        labels_d, centres_d = self._regions_device(prob_d, method)
        l_center_coordinates = centres_d.cpu().numpy()
        r_coordinates_segment = self._transform_layer_to_real(l_center_coordinates)
        return prob_d, l_center_coordinates, labels_d, raw_d, r_coordinates_segment

    def _predict_cellregions_device(self, raw_d, vol, read_cache=True):
        """reference :652-669 with the volume kept on the device; the float16 unet_cache file is read / written like upstream."""
        t = _dev.torch()
        cache = None if self.paths is None or self.paths.unet_cache is None else self.paths.unet_cache + "t%06i.npy" % vol
        if cache is not None and read_cache:
            try:
                a = np.load(cache, allow_pickle=True)
                return t.from_numpy(np.ascontiguousarray(a[0, :, :, :, 0], dtype=np.float32)).cuda()
            except OSError:
                pass
        from .preprocess import normalize_image_device
        if self.unet_model is None:
            raise ValueError("the 3D U-Net is not loaded: call load_unet() first")
        norm_d = normalize_image_device(raw_d, self.noise_level, (27, 27, 1), mode=0, subtract_median=True)
        prob_d = self.unet_model.predict_volume_device(norm_d, self.shrink)
        if cache is not None:
            np.save(cache, prob_d.cpu().numpy()[None, :, :, :, None].astype("float16"))
        return prob_d

    def _predict_cellregions(self, image_raw, vol):
        """reference :652-660 (numpy in / numpy [1, x, y, z, 1] out; a cache hit does not touch the image)."""
        cache = None if self.paths is None or self.paths.unet_cache is None else self.paths.unet_cache + "t%06i.npy" % vol
        if cache is not None:
            try:
                return np.load(cache, allow_pickle=True)
            except OSError:
                pass
        return self._save_unet_regions(image_raw, vol)

    def _save_unet_regions(self, image_raw, vol):
        """reference :662-669: _normalize_image -> unet3_prediction, cached as float16 [1, x, y, z, 1]."""
        t = _dev.torch()
        a = np.asarray(image_raw)
        raw_d = t.from_numpy(np.ascontiguousarray(a if a.dtype == np.uint16 else a.astype(np.float32))).cuda()
        return self._predict_cellregions_device(raw_d, vol, read_cache=False).cpu().numpy()[None, :, :, :, None]

    def _regions_device(self, prob_d, method):
        """_watershed (:671-684) + center_of_mass(regions > 0, regions, 1..n) (:646-647) on the device -> (labels, centres).
        region_method "watershed" (default): ct_watershed_segment = watershed_2d / watershed_3d with the reference's parameters
        (min_distance 7 / 3, sampling [1, 1, z_xy_ratio], method "min_size" | "cell_num"), relabel_sequential; self.min_size and
        self.cell_num are updated like :681-683.  region_method "cc": threshold 0.5 + connected components (touching cells stay one
        region)."""
        from .segment import segment_centroids_device, watershed_centroids_device
        t = _dev.torch()
        if float(prob_d.max()) <= 0.5:
            raise ValueError("No cell was detected by 3D U-Net! Try to reduce the noise_level.")
        if self.region_method == "watershed":
            if method not in ("min_size", "cell_num"):
                raise ValueError("The method parameter should be either min_size or cell_num")
            labels_d, centres_d, _, min_size, cell_num = watershed_centroids_device(prob_d, float(self.z_xy_ratio), method,
                                                                                    int(self.min_size or 0), int(self.cell_num or 0))
            if centres_d.shape[0] == 0:
                raise ValueError("No cell was detected by watershed! Try to reduce the min_size.")
            self.min_size = min_size
            if method == "min_size":
                self.cell_num = cell_num
            return labels_d, centres_d
        if self.region_method != "cc":
            raise ValueError(f"unknown region_method {self.region_method!r}: use 'watershed' or 'cc'")
        min_size = int(self.min_size or 0)
        labels_d, centres_d, sizes_d = segment_centroids_device(prob_d, 0.5, self.connectivity, min_size if method == "min_size" else 0)
        if method == "cell_num" and self.cell_num and centres_d.shape[0] > self.cell_num:
            # keep the cell_num largest regions (watershed_3d's "cell_num" method): the smallest kept size becomes min_size
            order = t.argsort(sizes_d, descending=True, stable=True)
            self.min_size = int(sizes_d[order[self.cell_num - 1]].item())
            labels_d, centres_d, sizes_d = segment_centroids_device(prob_d, 0.5, self.connectivity, self.min_size)
        if centres_d.shape[0] == 0:
            raise ValueError("No cell was detected by watershed! Try to reduce the min_size.")
        if method == "min_size":
            self.cell_num = int(centres_d.shape[0])
        return labels_d, centres_d

    def segment_prob(self, image_cell_bg, min_size=0, threshold=0.5, connectivity=1):
        """Regions and centres of a given probability map (numpy): (l_center_coordinates, segmentation_auto,
        r_coordinates_segment); records the segmentation for the next match."""
        from .segment import segment_centroids
        labels, l_centres, _ = segment_centroids(image_cell_bg, threshold, connectivity, min_size)
        r = self._transform_layer_to_real(l_centres)
        self.inject_segmentation(r, image_cell_bg=image_cell_bg)
        return l_centres, labels, r

    # ------------------------------------------------------------------ injection points (no file IO)
    def set_volume1(self, r_segmented_coordinates, r_tracked_coordinates=None):
        """State that segment_vol1 + interpolate_seg + initiate_tracking would leave, from coordinates the caller holds."""
        seg = np.asarray(r_segmented_coordinates, dtype=np.float64)
        trk = seg.copy() if r_tracked_coordinates is None else np.asarray(r_tracked_coordinates, dtype=np.float64)
        self.r_coordinates_segment_t0 = seg
        self.r_coordinates_tracked_t0 = trk
        self.cell_num_t0 = trk.shape[0]
        self.initiate_tracking(print_=False)

    def inject_segmentation(self, r_coordinates_segment, image_cell_bg=None, image_raw=None):
        """The target volume's segmentation from coordinates (and optionally prob map / raw stack) the caller holds."""
        t = _dev.torch()
        self._injected = True
        self.segresult.r_coordinates_segment = np.asarray(r_coordinates_segment, dtype=np.float64)
        if image_cell_bg is not None:
            a = np.asarray(image_cell_bg)
            self.segresult._image_cell_bg = None
            self.segresult.image_cell_bg_d = t.from_numpy(np.ascontiguousarray(a[0, :, :, :, 0] if a.ndim == 5 else a, dtype=np.float32)).cuda()
        if image_raw is not None:
            a = np.asarray(image_raw)
            self.segresult._image_gcn = None
            self.segresult.raw_d = t.from_numpy(np.ascontiguousarray(a if a.dtype == np.uint16 else a.astype(np.float32))).cuda()

    def set_interpolated_segmentation(self, seg_cells_interpolated_corrected):
        """What interpolate_seg (:1046-1075) leaves behind, from a label image on the z-interpolated grid the caller supplies
        (x, y, z * z_scaling; labels 1..n, cells not touching)."""
        from scipy import ndimage
        seg = np.asarray(seg_cells_interpolated_corrected)
        self.seg_cells_interpolated_corrected = seg
        self.Z_RANGE_INTERP = range(self.z_scaling // 2, seg.shape[2], self.z_scaling)
        self.segmentation_manual_relabels = seg[:, :, self.Z_RANGE_INTERP]
        lab = self.segmentation_manual_relabels
        centres = ndimage.center_of_mass(lab > 0, lab, range(1, int(lab.max()) + 1))
        r = self._transform_layer_to_real(centres)
        self.r_coordinates_tracked_t0 = r.copy()
        self.cell_num_t0 = r.shape[0]

    # ------------------------------------------------------------------ volume-1 bookkeeping
    def load_manual_seg(self):
        """reference :908-919 (PIL instead of tifffile; relabel_sequential restated with numpy)."""
        from PIL import Image
        folder = self.paths.manual_segmentation_vol1
        files = sorted(f for f in os.listdir(folder) if not f.startswith("."))
        seg = np.array([np.array(Image.open(os.path.join(folder, f))) for f in files]).transpose((1, 2, 0))
        present = np.unique(seg); present = present[present > 0]
        lut = np.zeros(int(seg.max()) + 1, dtype=np.int64); lut[present] = np.arange(1, present.size + 1)
        self.segmentation_manual_relabels = lut[seg]
        print("Loaded manual _segment at vol 1")

    def interpolate_seg(self):
        """reference :1046-1075.  The per-cell Gaussian smoothing (track.gaussian_filter :322-361; skimage.filters.gaussian ==
        scipy.ndimage.gaussian_filter(mode='constant', truncate=4)) runs on the host with scipy -- it is one-off volume-1 set-up,
        not the per-frame path.  Overlapping smoothed cells would need skimage's watershed (recalculate_cell_boundaries,
        watershed.py:111-151; absent, parity unpinned): NotImplementedError -- pass a finished label image to
        set_interpolated_segmentation() instead."""
        from scipy import ndimage
        img = np.asarray(self.segmentation_manual_relabels)
        zs = int(self.z_scaling)
        interp = np.repeat(img, zs, axis=2)
        out = np.zeros(tuple(s + 10 for s in interp.shape), dtype=int)
        mask = out.copy()
        for lab in range(1, int(img.max()) + 1):
            idx = np.where(interp == lab)
            lo = [int(a.min()) for a in idx]; hi = [int(a.max()) for a in idx]
            sub = np.zeros(tuple(hi[d] - lo[d] + 11 for d in range(3)))
            sub[idx[0] - lo[0] + 5, idx[1] - lo[1] + 5, idx[2] - lo[2] + 5] = 0.5
            percentage = 1 - np.divide(idx[0].size, sub.size, dtype="float")
            smooth = ndimage.gaussian_filter(sub, sigma=2.5, mode="constant", cval=0.0, truncate=4.0)
            region = smooth > np.percentile(smooth, percentage * 100)
            sl = tuple(slice(lo[d], hi[d] + 11) for d in range(3))
            out[sl] += region * lab
            mask[sl] += region * 1
        if (mask > 1).any():
            raise NotImplementedError("interpolate_seg: smoothed cells overlap; re-drawing their boundaries needs skimage's watershed "
                                      "(not available). Build the interpolated label image elsewhere and pass it to "
                                      "set_interpolated_segmentation().")
        seg = out[5:self.x_siz + 5, 5:self.y_siz + 5, 5:self.z_siz * zs + 5]
        # _relabel_separated_cells (:1077-1085): connected components per label value, full connectivity, raster order
        relab = np.zeros_like(seg); nxt = 0
        firsts = []
        for lab in np.unique(seg)[1:]:
            cc, n = ndimage.label(seg == lab, structure=np.ones((3, 3, 3)))
            for k in range(1, n + 1):
                firsts.append((int(np.flatnonzero(cc.ravel() == k)[0]), cc == k))
        for _, sel in sorted(firsts, key=lambda p: p[0]):
            nxt += 1; relab[sel] = nxt
        self.set_interpolated_segmentation(relab)

    def cal_subregions(self):
        """reference :1095-1112 (track.get_subregions :501-533) + upload of the packed sub-region masks."""
        from scipy import ndimage
        t = _dev.torch()
        seg_16 = self.seg_cells_interpolated_corrected.astype("int16")
        n = int(seg_16.max())
        self.region_list, self.region_width, self.region_xyz_min = [], [], []
        for lab, sl in enumerate(ndimage.find_objects(seg_16, max_label=n), start=1):
            if sl is None:
                raise ValueError(f"label {lab} is missing from the interpolated segmentation")
            self.region_list.append(seg_16[sl] == lab)
            self.region_width.append([s.stop - s.start for s in sl])
            self.region_xyz_min.append([s.start for s in sl])
        self.pad_x, self.pad_y, self.pad_z = (int(v) for v in np.max(self.region_width, axis=0))
        self.label_padding = None          # the padded work image of the reference lives on the device (overlap counts only)
        bbox = np.concatenate([np.asarray(self.region_xyz_min, dtype=np.int32), np.asarray(self.region_width, dtype=np.int32)], axis=1)
        chunks = [np.ascontiguousarray(r, dtype=np.uint8).ravel() for r in self.region_list]
        offs = np.concatenate([[0], np.cumsum([c.size for c in chunks])[:-1]]).astype(np.int64)
        self._regions_dev = (t.from_numpy(np.ascontiguousarray(bbox)).cuda(), t.from_numpy(np.concatenate(chunks)).cuda(),
                             t.from_numpy(offs).cuda())

    def _check_multicells(self):
        from scipy import ndimage
        for i, region in enumerate(self.region_list):
            assert ndimage.label(region)[1] == 1, f"more than one cell in region {i + 1}"

    def initiate_tracking(self, print_=True):
        """reference :1124-1136"""
        self.cells_on_boundary = np.zeros(self.cell_num_t0).astype(int)
        self.history.r_displacements = [np.zeros((self.cell_num_t0, 3))]
        self.history.r_segmented_coordinates = [self.r_coordinates_segment_t0]
        self.history.r_tracked_coordinates = [self.r_coordinates_tracked_t0]
        self.history.anim = []
        if print_:
            print("Initiated coordinates for tracking (from vol 1)")

    # ------------------------------------------------------------------ matching
    def match(self, target_volume, method="min_size"):
        """reference :1138-1175 -> (anim, [cells_on_boundary_local, target_volume, i_disp_from_vol1_updated, r_coor_predicted]).
        anim is None (drawing is not part of this path).  If a segmentation was injected (inject_segmentation) and no image
        source is configured, it is used instead of _segment."""
        if target_volume in self.miss_frame:
            raise ValueError("target_volume is a miss_frame")
        if self._injected:
            self._injected = False
        elif self.image_reader is not None or (self.paths is not None and self.paths.raw_image is not None):
            self.segresult.update_results(*self._segment(target_volume, method=method))
        else:
            raise ValueError("no image source and no injected segmentation for the target volume")
        r_coor_predicted, anim = self._predict_pos_once(source_volume=1, draw=False)
        cells_on_boundary_local = self.cells_on_boundary.copy()
        if self.x_siz is not None:        # `a[()] = 1` would flag EVERY cell: without a volume size there is no boundary test
            cells_on_boundary_local[self._get_cells_onBoundary(r_coor_predicted, self.ensemble)] = 1
        i_disp_from_vol1_updated = None
        if self._regions_dev is not None and self.segresult.image_cell_bg_d is not None:
            _, i_disp_from_vol1_updated = self._accurate_correction(cells_on_boundary_local, r_coor_predicted)
        return anim, [cells_on_boundary_local, target_volume, i_disp_from_vol1_updated, r_coor_predicted]

    def _accurate_correction(self, cells_on_boundary_local, r_coor_predicted):
        """reference :1177-1191 -> (r_disp_from_vol1_updated, i_disp_from_vol1_updated), on the device."""
        t = _dev.torch(); L = _lib.lib()
        if self._regions_dev is None:
            raise ValueError("cal_subregions() has not been called")
        prob_d = self.segresult.image_cell_bg_d
        if prob_d is None:
            raise ValueError("no probability map for the target volume")
        raw_d = self.segresult.raw_d
        n = int(self.cell_num_t0)
        r_disp0 = self.history.r_displacements[-1] + (np.asarray(r_coor_predicted) - self.history.r_tracked_coordinates[-1])
        r_disp_d = _dev.to_dev(r_disp0, t.float64, prob_d.device)
        i_disp_d = _dev.empty((n, 3), t.int32, prob_d.device)
        bd_d = t.from_numpy(np.ascontiguousarray(np.asarray(cells_on_boundary_local) != 0, dtype=np.uint8)).to(prob_d.device)
        t0_d = _dev.to_dev(self.r_coordinates_tracked_t0, t.float64, prob_d.device)
        bbox_d, subs_d, offs_d = self._regions_dev
        dims = _lib.ivec(prob_d.shape)
        ws = _dev.workspace(L.ct_correction_legacy_workspace_bytes(dims, n), prob_d.device)
        iters = C.c_int(0)
        raw_dtype = 0 if (raw_d is not None and raw_d.dtype == t.uint16) else 1
        if raw_d is not None and raw_dtype == 1 and raw_d.dtype != t.float32:
            raw_d = raw_d.to(t.float32)
        _lib.check(L.ct_accurate_correction_legacy(
            prob_d.data_ptr(), raw_d.data_ptr() if raw_d is not None else None, raw_dtype, dims, int(self.z_scaling),
            int(self.seg_cells_interpolated_corrected.shape[2]), float(self.z_xy_ratio), n, bbox_d.data_ptr(), subs_d.data_ptr(),
            offs_d.data_ptr(), _lib.ivec((self.pad_x, self.pad_y, self.pad_z)), bd_d.data_ptr(), t0_d.data_ptr(), r_disp_d.data_ptr(),
            i_disp_d.data_ptr(), REP_NUM_CORRECTION, C.byref(iters), ws.data_ptr(), ws.numel(), _dev.stream(prob_d.device)),
            "ct_accurate_correction_legacy")
        self.last_correction_rounds = iters.value
        return r_disp_d.cpu().numpy(), i_disp_d.cpu().numpy().astype(int)

    def _fit_device(self, seg_pre_d, seg_tgt_d, rep):
        C_t, beta_t, inter_t = [], [], []
        inter = seg_pre_d
        _dev.check_match_sizes(seg_pre_d.shape[0], seg_tgt_d.shape[0], 20, "Tracker._fit_ffn_prgls")
        for i in range(rep):
            beta = self.beta_tk * (0.8 ** i)
            inter_t.append(inter)
            if isinstance(self.ffn_model, FFN):
                corr = initial_matching_device(self.ffn_model, inter, seg_tgt_d, 20)
            else:
                corr = _dev.to_dev(initial_matching_quick(self.ffn_model, inter.cpu().numpy(), seg_tgt_d.cpu().numpy(), 20),
                                   _dev.torch().float32)
            _, moved, Cm = _dev.prgls_legacy(inter, seg_tgt_d, corr, beta, self.max_iteration, self.lambda_tk, 1e8, want_P=False)
            inter = moved
            C_t.append(Cm); beta_t.append(beta)
        return C_t, beta_t, inter_t

    def _predict_pos_device(self, source_volume):
        seg_pre = _dev.points_dev(self.history.r_segmented_coordinates[source_volume - 1])
        seg_tgt = _dev.points_dev(self.segresult.r_coordinates_segment)
        tracked_pre = _dev.points_dev(self.history.r_tracked_coordinates[source_volume - 1])
        if isinstance(self.ffn_model, FFN) and self.ffn_model._handle is not None:
            # the whole chain (5 x [features, FFN, PR-GLS] + 5 x Gram application) as one native call: the threads that drive the
            # independent source volumes of an ensemble prediction then run without the interpreter lock
            _dev.check_match_sizes(seg_pre.shape[0], seg_tgt.shape[0], 20, "Tracker._fit_ffn_prgls")
            return _dev.legacy_predict_pos(self.ffn_model._handle, seg_pre, seg_tgt, tracked_pre, self.beta_tk, self.lambda_tk,
                                           self.max_iteration, REP_NUM_PRGLS, 20)
        return self._predict_pos_composed(seg_pre, seg_tgt, tracked_pre)

    def _predict_pos_composed(self, seg_pre, seg_tgt, tracked_pre):
        """The same chain call by call (foreign FFN objects; the parity test of the fused entry point)."""
        C_t, beta_t, inter_t = self._fit_device(seg_pre, seg_tgt, REP_NUM_PRGLS)
        pred = tracked_pre.clone()
        for Cm, b, inter in zip(C_t, beta_t, inter_t):
            _dev.gram_apply(pred, inter, Cm, b)
        return pred

    def _predict_pos_once(self, source_volume, draw=False):
        """reference :1193-1222 (the animation of the draw=True branch is not produced: anim is None)."""
        return self._predict_pos_device(source_volume).cpu().numpy(), None

    def _fit_ffn_prgls(self, rep, r_coordinates_segment_pre):
        """reference :1224-1254 -> (C_t, BETA_t, coor_intermediate_list) as numpy."""
        C_t, beta_t, inter_t = self._fit_device(_dev.points_dev(r_coordinates_segment_pre),
                                                _dev.points_dev(self.segresult.r_coordinates_segment), rep)
        return [c.cpu().numpy() for c in C_t], beta_t, [x.cpu().numpy() for x in inter_t]

    def _ffn_prgls_once(self, i, r_coordinates_segment_pre):
        """reference :1256-1267"""
        init_match = initial_matching_quick(self.ffn_model, r_coordinates_segment_pre, self.segresult.r_coordinates_segment, 20)
        P, post, Cm = pr_gls_quick(np.array(r_coordinates_segment_pre, copy=True), self.segresult.r_coordinates_segment,
                                   init_match, BETA=self.beta_tk * (0.8 ** i), max_iteration=self.max_iteration, LAMBDA=self.lambda_tk)
        return Cm, post

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
        """Ensemble part of track_one_vol (reference :1499-1509): every source volume's prediction, trim-mean'd.  The source
        volumes are independent: sharded over the ranks when torch.distributed is initialised, `ensemble_chains` at a time
        inside a rank.  `history` must hold all source volumes."""
        from . import parallel
        t = _dev.torch()
        vols = get_reference_vols(self.ensemble, vol, adjacent=self.adjacent) if source_vols is None else source_vols
        stack = parallel.sharded_map_gather(lambda v: self._predict_pos_device(v), vols, tail_shape=(int(self.cell_num_t0), 3),
                                            dtype=t.float64, chains=self.ensemble_chains, batch_fn=self._ensemble_batch_fn())
        return _dev.trim_mean(stack, 0.1).cpu().numpy()

    def _ensemble_batch_fn(self):
        """This rank's source volumes as ONE chain of launches (ct_legacy_predict_pos_batched) when the native FFN is in use and every
        point set is small enough for the single-workgroup dense M-step; None -> concurrent per-volume chains."""
        if not (isinstance(self.ffn_model, FFN) and self.ffn_model._handle is not None and self.ensemble_batched):
            return None
        seg_tgt_np = self.segresult.r_coordinates_segment
        sizes = [len(x) for x in self.history.r_segmented_coordinates] + [len(seg_tgt_np)]
        if not sizes or max(sizes) > _dev.LEGACY_BATCH_MAX_POINTS or min(sizes) <= 20:
            return None

        def run(vols):
            seg_tgt = _dev.points_dev(seg_tgt_np)
            pre = [_dev.points_dev(self.history.r_segmented_coordinates[v - 1]) for v in vols]
            trk = [_dev.points_dev(self.history.r_tracked_coordinates[v - 1]) for v in vols]
            out = _dev.legacy_predict_pos_batched(self.ffn_model._handle, pre, seg_tgt, trk, self.beta_tk, self.lambda_tk, self.max_iteration,
                                                  REP_NUM_PRGLS, 20)
            return [out[i] for i in range(len(vols))]
        return run

    # ------------------------------------------------------------------ tracking loop (reference :1415-1551)
    def track(self, fig=None, ax=None, from_volume=2):
        self._reset_tracking_state(from_volume)
        for vol in range(from_volume, self.volume_num + 1):
            self.track_one_vol(vol, fig, ax)
        return None

    def _reset_tracking_state(self, from_volume):
        """Forget everything recorded from `from_volume` on, so that tracking can restart there (reference :1462-1470: the same two
        conditions and messages)."""
        assert from_volume >= 2, "from_volume should >= 2"
        hist, keep = self.history, from_volume - 1
        tracked_until = len(hist.r_displacements)
        for series in (hist.r_displacements, hist.r_segmented_coordinates, hist.r_tracked_coordinates):
            series[keep:] = []
        assert len(hist.r_displacements) == keep, \
            f"Currently data has been tracked until vol {tracked_until}, the program cannot start from {from_volume}"

    def track_one_vol(self, target_volume, fig=None, axc6=None, method="min_size"):
        """reference :1473-1536 without the label-image / figure outputs (skimage watershed, matplotlib)."""
        if target_volume in self.miss_frame:
            self.history.r_displacements.append(self.history.r_displacements[-1])
            self.history.r_segmented_coordinates.append(self.segresult.r_coordinates_segment)
            self.history.r_tracked_coordinates.append(self.r_coordinates_tracked_t0 + self.history.r_displacements[-1])
            return None
        self.segresult.update_results(*self._segment(target_volume, method=method))
        r_coor_predicted_mean = self.predict_ensemble(target_volume)
        cells_bd = self._get_cells_onBoundary(r_coor_predicted_mean, self.ensemble)
        self.cells_on_boundary[cells_bd] = 1
        r_disp_from_vol1_updated, i_disp_from_vol1_updated = self._accurate_correction(self.cells_on_boundary, r_coor_predicted_mean)
        self.last_i_disp = i_disp_from_vol1_updated
        if self.ensemble:
            self.cells_on_boundary = np.zeros(self.cell_num_t0).astype(int)
        self.history.r_displacements.append(r_disp_from_vol1_updated)
        self.history.r_segmented_coordinates.append(self.segresult.r_coordinates_segment)
        self.history.r_tracked_coordinates.append(self.r_coordinates_tracked_t0 + r_disp_from_vol1_updated)
        return None

    def save_coordinates(self):
        """reference :1538-1551"""
        coord = np.asarray(self.history.r_tracked_coordinates)
        t, cell, pos = coord.shape
        coord_table = np.column_stack((np.repeat(np.arange(1, t + 1), cell), np.tile(np.arange(1, cell + 1), t),
                                       coord.reshape(t * cell, pos)))
        np.savetxt(os.path.join(self.paths.track_information, "tracked_coordinates.csv"), coord_table, delimiter=",",
                   header="cell,t,x(row),y(column),z(interpolated)", comments="")


for _name in ("draw_segresult", "draw_manual_seg1", "draw_correction", "draw_overlapping", "subplots_tracking", "replay_track_animation",
              "retrain_unet", "select_unet_weights", "_transform_motion_to_image"):
    setattr(Tracker, _name, _not_accelerated(_name))
