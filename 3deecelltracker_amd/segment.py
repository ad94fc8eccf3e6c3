"""Probability map -> labelled regions -> cell centres on the GPU.

Reference: ``Tracker._segment`` (CellTracker/tracker.py:636-648) = U-Net prob map -> ``_watershed`` (:671-684, skimage
marker watershed in watershed.py:16-108) -> ``scipy.ndimage.center_of_mass(regions > 0, regions, 1..n)`` ->
``_transform_layer_to_real``.

Two region steps, both on the device:

* ``watershed_centroids[_device]`` -- the reference's own marker watershed (watershed_2d per z slice, then watershed_3d, min_size /
  cell_num, relabel_sequential; ct_watershed_segment).  Held to the reference's own code running on scikit-image 0.18.3
  (tests/golden/watershed_skimage.npz, tests/test_watershed_pin.py: eight volumes incl. the 512 x 512 x 32 benchmark stack, voxel for
  voxel, incl. numpy's sort order among exactly tied peak candidates);
  scipy's EDT / Gaussian arithmetic is reproduced operand for operand.  This is what ``Tracker._segment`` uses.
* ``segment_centroids[_device]`` -- threshold + 3D connected components (touching cells are not split): the cheap variant SURVEY 8f#2
  names, kept as ``method="cc"`` and used by the per-frame ``FrameChain``.

Label numbering (raster order, small regions removed, renumbered 1..n) and the centre-of-mass call follow the reference, so the centres
feed ``Tracker.match`` / ``TrackerLite`` exactly like ``seg/coords%06d.npy`` does (raw voxel coordinates, float64).
"""
from __future__ import annotations

import numpy as np

from . import _dev, _lib


def segment_centroids_device(prob, threshold: float = 0.5, connectivity: int = 1, min_size: int = 0, cap: int = 4096,
                             want_labels: bool = True):
    """prob: contiguous float32 cuda tensor [x, y, z] -> (labels int32 cuda | None, centres fp64 cuda [n, 3], sizes int32 cuda [n]).

    `cap` is the initial capacity of the centre table; it is doubled and the call repeated when the volume holds more regions."""
    t = _dev.torch(); L = _lib.lib()
    if prob.dim() != 3 or not prob.is_cuda or not prob.is_contiguous() or prob.dtype != t.float32:
        raise ValueError("expected a contiguous float32 cuda tensor (x, y, z)")
    if connectivity not in (1, 2, 3):
        raise ValueError("connectivity must be 1, 2 or 3")
    if min_size < 0 or cap <= 0:
        raise ValueError("min_size must be >= 0 and cap positive")
    dims = _lib.ivec(prob.shape)
    labels = _dev.empty(tuple(prob.shape), t.int32, prob.device) if want_labels else None
    n_dev = _dev.empty((1,), t.int32, prob.device)
    while True:
        centres = _dev.empty((cap, 3), t.float64, prob.device)
        sizes = _dev.empty((cap,), t.int32, prob.device)
        nbytes = L.ct_segment_workspace_bytes(dims, int(cap))
        if nbytes == 0:
            raise ValueError(f"volume {tuple(prob.shape)} is too large for int32 voxel indices")
        ws = _dev.workspace(nbytes, prob.device)
        _lib.check(L.ct_segment_centroids(prob.data_ptr(), dims, float(threshold), int(connectivity), int(min_size),
                                          int(cap), labels.data_ptr() if want_labels else None, centres.data_ptr(),
                                          sizes.data_ptr(), n_dev.data_ptr(), ws.data_ptr(), ws.numel(),
                                          _dev.stream(prob.device)), "ct_segment_centroids")
        n = int(n_dev.item())
        if n <= cap:
            return labels, centres[:n], sizes[:n]
        cap = max(2 * cap, n)


def segment_centroids(prob, threshold: float = 0.5, connectivity: int = 1, min_size: int = 0):
    """numpy (x, y, z) prob map -> (labels int32, centres float64 [n, 3], sizes int32 [n]).

    Raises ValueError("No cell was detected ...") like tracker.py:637-643 when nothing exceeds the threshold / survives."""
    t = _dev.torch()
    a = np.asarray(prob)
    if a.ndim == 5:                       # the reference passes unet3_prediction's [1, x, y, z, 1]
        a = a[0, :, :, :, 0]
    if a.ndim != 3:
        raise ValueError(f"expected a 3-D probability map (x, y, z), got shape {a.shape}")
    d = t.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
    labels, centres, sizes = segment_centroids_device(d, threshold, connectivity, min_size)
    if centres.shape[0] == 0:
        raise ValueError("No cell was detected! Try to reduce the min_size / noise_level.")
    return labels.cpu().numpy(), centres.cpu().numpy(), sizes.cpu().numpy()


def gaussian_weights(sigma: float, truncate: float = 4.0):
    """The correlation weights scipy.ndimage.gaussian_filter1d builds for `sigma` (order 0): -> (float64 [2 r + 1], r)."""
    sd = float(sigma)
    radius = int(truncate * sd + 0.5)
    x = np.arange(-radius, radius + 1)
    phi = np.exp(-0.5 / (sd * sd) * x ** 2)
    phi = phi / phi.sum()
    return np.ascontiguousarray(phi[::-1], dtype=np.float64), radius


_METHODS = {"min_size": 0, "cell_num": 1}
# what ct_watershed_workspace_bytes_ex / ct_watershed_segment_ex refuse (csrc/ct_segment.hip): the peak tables are sized per call and grown on
# overflow, the per-slice statistics from the z extent -- what is left is the index arithmetic
WATERSHED_LIMITS = "x, y and z < 16384 and < 2^31 voxels"
PEAK_CAP_2D, PEAK_CAP_3D = 2048, 8192          # peak-candidate slots per z slice / in the volume of the first attempt (every stack measured fits)
PEAK_CAP_MAX_2D, PEAK_CAP_MAX_3D = 1 << 22, 1 << 24


class PendingWatershed:
    """ct_watershed_segment_ex enqueued on a stream (watershed_centroids_enqueue); result() makes the call's only host round trip (the region
    count), re-running with larger tables in the rare cases that `cap` regions or the peak-candidate tables were not enough (the reference's
    watershed.py takes any stack: the device's tables grow to what the stack needs)."""

    def __init__(self, **kw):
        self.peak_cap_2d, self.peak_cap_3d = PEAK_CAP_2D, PEAK_CAP_3D
        self.retries = 0
        self.__dict__.update(kw)

    def _enqueue(self):
        import ctypes as C
        t = _dev.torch(); L = _lib.lib()
        prob, cap = self.prob, self.cap
        dims = _lib.ivec(prob.shape)
        self.centres = _dev.empty((cap, 3), t.float64, prob.device)
        self.sizes = _dev.empty((cap,), t.int32, prob.device)
        nbytes = L.ct_watershed_workspace_bytes_ex(dims, int(cap), int(self.peak_cap_2d), int(self.peak_cap_3d))
        if nbytes == 0:
            raise ValueError(f"volume {tuple(prob.shape)} is outside the device watershed's limits ({WATERSHED_LIMITS}); "
                             "threshold + connected components have none: Tracker.region_method = 'cc' / segment_centroids_device")
        self.ws = _dev.workspace(nbytes, prob.device)
        rc = L.ct_watershed_segment_ex(prob.data_ptr(), dims, float(self.z_xy_ratio), _METHODS[self.method], int(self.min_size), int(self.cell_num),
                                       int(self.min_distance_2d), int(self.min_distance_3d), self.w_xy.ctypes.data_as(C.c_void_p), self.r_xy,
                                       self.w_z.ctypes.data_as(C.c_void_p), self.r_z, int(cap), int(self.peak_cap_2d), int(self.peak_cap_3d),
                                       self.labels.data_ptr() if self.labels is not None else None,
                                       self.centres.data_ptr(), self.sizes.data_ptr(), self.n_dev.data_ptr(), self.ws.data_ptr(), self.ws.numel(),
                                       _dev.stream(prob.device))
        _lib.check(rc, "ct_watershed_segment_ex")
        self.stream = t.cuda.current_stream(prob.device)
        self.event = t.cuda.Event(); self.event.record(self.stream)

    def result(self):
        """-> (labels int32 cuda | None, centres fp64 cuda [n, 3], sizes int32 cuda [n], min_size in force, cell_num in force); usable on the
        stream that is current when result() is called."""
        t = _dev.torch()
        while True:
            self.event.synchronize()
            cur = t.cuda.current_stream(self.prob.device)
            if cur != self.stream:
                for x in (self.centres, self.sizes, self.n_dev, self.labels):
                    if x is not None:
                        x.record_stream(cur)
            n, ms, cn = (int(v) for v in self.n_dev.cpu().tolist())          # the call's only host round trip
            if n == -2:
                # a peak table overflowed in one of the stages (latched on the device, nothing was waited for); ms / cn = the slots the fullest z
                # slice / the volume wanted.  The 3-D stage only ran on garbage if the 2-D stage overflowed, so its figure counts only when the
                # 2-D tables held; a second overflow (the 3-D stage after a 2-D retry) takes one more round.
                want2, want3 = ms, cn
                grow2 = want2 > self.peak_cap_2d
                new2 = max(self.peak_cap_2d, _pow2_ceil(want2)) if grow2 else self.peak_cap_2d
                new3 = max(self.peak_cap_3d, _pow2_ceil(want3)) if (want3 > self.peak_cap_3d and not grow2) else self.peak_cap_3d
                if grow2 and want3 > self.peak_cap_3d:
                    new3 = max(self.peak_cap_3d, _pow2_ceil(min(want3, PEAK_CAP_MAX_3D)))      # (a lower bound taken from the garbage run: saves a round when it holds)
                if (new2, new3) == (self.peak_cap_2d, self.peak_cap_3d) or new2 > PEAK_CAP_MAX_2D or new3 > PEAK_CAP_MAX_3D or self.retries >= 4:
                    raise ValueError(f"the probability map has more peak candidates ({want2} in a z slice / {want3} in the volume) than the device "
                                     f"watershed's tables can be grown to ({PEAK_CAP_MAX_2D} / {PEAK_CAP_MAX_3D}): a map that noisy usually needs a higher "
                                     "noise_level; Tracker.region_method = 'cc' has no such limit")
                self.peak_cap_2d, self.peak_cap_3d = new2, new3
                self.retries += 1
                with t.cuda.stream(self.stream):
                    self._enqueue()
                continue
            if n < 0:                         # watershed.py:92: np.sort(counts)[-cell_num - 1] with fewer than cell_num + 1 bins
                raise IndexError(f"index {-self.cell_num - 1} is out of bounds: method='cell_num' asks for {self.cell_num} cells, the watershed found fewer regions")
            if n <= self.cap:
                return self.labels, self.centres[:n], self.sizes[:n], ms, cn
            self.cap = max(2 * self.cap, n)
            with t.cuda.stream(self.stream):
                self._enqueue()


def _pow2_ceil(v: int) -> int:
    p = 16
    while p < v:
        p <<= 1
    return p


def watershed_centroids_enqueue(prob, z_xy_ratio: float, method: str = "min_size", min_size: int = 0, cell_num: int = 0, cap: int = 4096,
                                want_labels: bool = True, min_distance_2d: int = 7, min_distance_3d: int = 3,
                                peak_cap_2d: int = PEAK_CAP_2D, peak_cap_3d: int = PEAK_CAP_3D) -> PendingWatershed:
    """watershed_centroids_device without its host round trip: the kernels are enqueued on the current stream, `.result()` waits for them.
    `prob` must stay untouched until result() has returned."""
    t = _dev.torch()
    if prob.dim() != 3 or not prob.is_cuda or not prob.is_contiguous() or prob.dtype != t.float32:
        raise ValueError("expected a contiguous float32 cuda tensor (x, y, z)")
    if method not in _METHODS:
        raise ValueError("The method parameter should be either min_size or cell_num")        # watershed.py:93
    if min_size < 0 or cell_num < 0 or cap <= 0:
        raise ValueError("min_size / cell_num must be >= 0 and cap positive")
    w_xy, r_xy = gaussian_weights(2.0)
    w_z, r_z = gaussian_weights(0.3)
    p = PendingWatershed(prob=prob, z_xy_ratio=z_xy_ratio, method=method, min_size=min_size, cell_num=cell_num, cap=cap,
                         min_distance_2d=min_distance_2d, min_distance_3d=min_distance_3d, w_xy=w_xy, r_xy=r_xy, w_z=w_z, r_z=r_z,
                         labels=_dev.empty(tuple(prob.shape), t.int32, prob.device) if want_labels else None,
                         n_dev=_dev.empty((3,), t.int32, prob.device), peak_cap_2d=int(peak_cap_2d), peak_cap_3d=int(peak_cap_3d))
    p._enqueue()
    return p


def watershed_centroids_device(prob, z_xy_ratio: float, method: str = "min_size", min_size: int = 0, cell_num: int = 0, cap: int = 4096,
                               want_labels: bool = True, min_distance_2d: int = 7, min_distance_3d: int = 3,
                               peak_cap_2d: int = PEAK_CAP_2D, peak_cap_3d: int = PEAK_CAP_3D):
    """Tracker._watershed (reference tracker.py:671-684 = watershed.py:16-108) + relabel_sequential + center_of_mass on the device.
    prob: contiguous float32 cuda tensor [x, y, z] -> (labels int32 cuda | None, centres fp64 cuda [n, 3], sizes int32 cuda [n],
    min_size in force, cell_num in force)."""
    return watershed_centroids_enqueue(prob, z_xy_ratio, method, min_size, cell_num, cap, want_labels, min_distance_2d, min_distance_3d,
                                       peak_cap_2d, peak_cap_3d).result()


def watershed_centroids(prob, z_xy_ratio: float, method: str = "min_size", min_size: int = 0, cell_num: int = 0):
    """numpy (x, y, z) or (1, x, y, z, 1) prob map -> (segmentation_auto int32, centres float64 [n, 3], min_size, cell_num)."""
    t = _dev.torch()
    a = np.asarray(prob)
    if a.ndim == 5:
        a = a[0, :, :, :, 0]
    if a.ndim != 3:
        raise ValueError(f"expected a 3-D probability map (x, y, z), got shape {a.shape}")
    d = t.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
    labels, centres, _, ms, cn = watershed_centroids_device(d, z_xy_ratio, method, min_size, cell_num)
    return labels.cpu().numpy(), centres.cpu().numpy(), ms, cn


def watershed_stages_device(prob, z_xy_ratio: float, stage: str = "2d", min_size: int = 0, cap: int = 4096):
    """Intermediates of the device watershed for the tests (ct_watershed_read_stage): stage "2d" = after watershed_2d (per-slice EDT,
    smoothed EDT, window maximum, per-slice watershed labels, the two masks), "3d" = after the whole call (the 3-D stage's smoothed EDT,
    window maximum and watershed labels before relabelling; the EDT array is scratch by then).  -> dict of numpy arrays."""
    import ctypes as C
    t = _dev.torch(); L = _lib.lib()
    dims = _lib.ivec(prob.shape)
    w_xy, r_xy = gaussian_weights(2.0)
    w_z, r_z = gaussian_weights(0.3)
    centres = _dev.empty((cap, 3), t.float64, prob.device); sizes = _dev.empty((cap,), t.int32, prob.device)
    n_dev = _dev.empty((3,), t.int32, prob.device)
    ws = _dev.workspace(L.ct_watershed_workspace_bytes(dims, int(cap)), prob.device)
    _lib.check(L.ct_watershed_segment(prob.data_ptr(), dims, float(z_xy_ratio), 0x100 if stage == "2d" else 0x200, int(min_size), 0, 7, 3,
                                      w_xy.ctypes.data_as(C.c_void_p), r_xy, w_z.ctypes.data_as(C.c_void_p), r_z, int(cap), None,
                                      centres.data_ptr(), sizes.data_ptr(), n_dev.data_ptr(), ws.data_ptr(), ws.numel(),
                                      _dev.stream(prob.device)), "ct_watershed_segment")
    out = {}
    for name, which, dt in (("mask", 0, t.uint8), ("mask_wo_boundaries", 1, t.uint8), ("edt", 2, t.float64), ("smooth", 3, t.float64),
                            ("labels", 4, t.int32), ("window_max", 5, t.float64)):
        buf = _dev.empty(tuple(prob.shape), dt, prob.device)
        _lib.check(L.ct_watershed_read_stage(ws.data_ptr(), dims, int(cap), which, buf.data_ptr(), _dev.stream(prob.device)), "ct_watershed_read_stage")
        out[name] = buf.cpu().numpy()
    return out
