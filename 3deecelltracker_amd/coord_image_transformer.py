"""`Coordinates` API type (reference CellTracker/coord_image_transformer.py:29-141).

Host-side value type only: stores raw voxel coordinates as float32 and exposes the real /
interpolated / integer views.  The label-image bookkeeping of the reference module
(CoordsToImageTransformer, plotting, TIFF IO) is outside the accelerated path.
"""
from __future__ import annotations

import numpy as np


class Coordinates:
    def __init__(self, coords, interpolation_factor: int, voxel_size, dtype: str = "raw"):
        self.interpolation_factor = interpolation_factor
        self.voxel_size = np.asarray(voxel_size)
        c32 = np.asarray(coords).astype(np.float32)
        if dtype == "raw":
            self._raw = c32
        elif dtype == "real":
            self._raw = self._scale(c32, 1.0 / self.voxel_size).astype(np.float32)
        elif dtype == "interp":
            self._raw = self._scale(c32, np.asarray((1, 1, 1 / interpolation_factor))).astype(np.float32)
        else:
            raise ValueError(f"dtype must be 'raw', 'real' or 'interp', got {dtype!r}")

    @staticmethod
    def _scale(coords_nx3, factor_x3):
        return coords_nx3 * np.asarray(factor_x3)[None, :]

    def __add__(self, other: "Coordinates") -> "Coordinates":
        return Coordinates(self._raw + other._raw, self.interpolation_factor, self.voxel_size, "raw")

    def __sub__(self, other: "Coordinates") -> "Coordinates":
        return Coordinates(self._raw - other._raw, self.interpolation_factor, self.voxel_size, "raw")

    @property
    def real(self) -> np.ndarray:
        return self._scale(self._raw, self.voxel_size)

    @property
    def interp(self) -> np.ndarray:
        return np.round(self._scale(self._raw, np.asarray((1, 1, self.interpolation_factor)))).astype(np.int32)

    @property
    def raw(self) -> np.ndarray:
        return np.round(self._raw).astype(np.int32)

    @property
    def cell_num(self) -> int:
        return self._raw.shape[0]


class CoordsToImageTransformer:
    """The accurate-correction part of the reference's CoordsToImageTransformer (coord_image_transformer.py:143-489).

    Only what the correction loop needs is kept: the per-cell sub-regions on the z-interpolated grid (the reference builds
    them with skimage in `interpolate`; here they are supplied as (bbox slices, mask) pairs), the volume-1 centres and the
    image geometry.  `accurate_correction` returns the corrected Coordinates; the corrected *label image* of the reference
    goes through skimage's watershed and is outside this path."""

    def __init__(self, proofed_shape, voxel_size, interpolation_factor, subregions, coord_vol1: Coordinates,
                 results_folder=None):
        from pathlib import Path
        self.results_folder = Path(results_folder) if results_folder is not None else None
        self.proofed_shape = tuple(int(v) for v in proofed_shape)
        self.voxel_size = np.asarray(voxel_size)
        self.interpolation_factor = int(interpolation_factor)
        self.subregions = list(subregions)
        self.coord_vol1 = coord_vol1
        self.z_slice_original_labels = slice(self.interpolation_factor // 2,
                                             self.interpolation_factor * self.proofed_shape[2], self.interpolation_factor)
        self._dev = None

    def get_cells_on_boundary(self, coordinates_real_nx3, ensemble: bool, boundary_xy: int = 6):
        """reference :371-404"""
        if ensemble:
            boundary_xy = 0
        x_siz, y_siz, z_siz = self.proofed_shape
        x, y, z = np.asarray(coordinates_real_nx3).T
        near = ((x < boundary_xy) | (y < boundary_xy) | (x > (x_siz - boundary_xy) * self.voxel_size[0]) |
                (y > (y_siz - boundary_xy) * self.voxel_size[1]) | (z < 0) | (z > z_siz * self.voxel_size[2]))
        return np.where(near)[0] + 1

    def _upload(self):
        from . import _dev
        t = _dev.torch()
        n = len(self.subregions)
        bbox = np.zeros((n, 6), dtype=np.int32)
        offs = np.zeros(n, dtype=np.int64)
        chunks = []
        pos = 0
        for i, (bb, sub) in enumerate(self.subregions):
            sub = np.ascontiguousarray(np.asarray(sub) != 0, dtype=np.uint8)
            bbox[i, :3] = [s.start for s in bb]
            bbox[i, 3:] = sub.shape
            if tuple(s.stop - s.start for s in bb) != sub.shape:
                raise ValueError("sub-image shape does not match its bounding box")
            offs[i] = pos; pos += sub.size
            chunks.append(sub.ravel())
        self._dev = (t.from_numpy(bbox).cuda(), t.from_numpy(np.concatenate(chunks)).cuda(), t.from_numpy(offs).cuda(),
                     t.from_numpy(np.ascontiguousarray(self.coord_vol1._raw, dtype=np.float32)).cuda())

    # ---- on-disk layout of the reference: results_folder/seg/prob%06d.npy, track_results/coords_real/coords%06d.npy
    def save_coords_vol1(self, t_start: int):
        """reference :265-267: the confirmed coordinates of the first volume."""
        path = self._coords_real_dir()
        path.mkdir(parents=True, exist_ok=True)
        np.save(str(path / ("coords%06d.npy" % t_start)), self.coord_vol1.real)

    def save_coords(self, t2: int, coords: Coordinates):
        """the coordinate part of reference :512 (label images / figures are outside this path)."""
        path = self._coords_real_dir()
        path.mkdir(parents=True, exist_ok=True)
        np.save(str(path / ("coords%06d.npy" % t2)), coords.real)

    def load_confirmed_coords(self, t1: int) -> np.ndarray:
        """reference :516-517"""
        return np.load(str(self._coords_real_dir() / f"coords{str(t1).zfill(6)}.npy"))

    def _coords_real_dir(self):
        if self.results_folder is None:
            raise ValueError("results_folder was not given to CoordsToImageTransformer")
        return self.results_folder / "track_results" / "coords_real"

    def accurate_correction(self, t, *args, **kwargs):
        """reference :406-447.  Two call forms:

        accurate_correction(t: int, grid, coords, ensemble, max_repetition=20, format="prob%06d.npy")   -- the reference's:
            loads results_folder/seg/prob%06d.npy, returns (coords, corrected_labels_image) with corrected_labels_image = None
            (the reference rebuilds the label image with a skimage watershed, which is outside this path);
        accurate_correction(prob_map, coords, ensemble, max_repetition=20, grid=(1, 1, 1))              -- array / cuda tensor in,
            Coordinates out (no file access)."""
        from_file = isinstance(t, (int, np.integer))
        names = ("grid", "coords", "ensemble", "max_repetition", "format") if from_file else ("coords", "ensemble", "max_repetition", "grid")
        if len(args) > len(names):
            raise TypeError("accurate_correction: too many positional arguments")
        params = dict(zip(names, args))
        for k, v in kwargs.items():
            if k not in names or k in params:
                raise TypeError(f"accurate_correction: unexpected or repeated argument '{k}'")
            params[k] = v
        for need in ("coords", "ensemble") + (("grid",) if from_file else ()):
            if need not in params:
                raise TypeError(f"accurate_correction: missing argument '{need}'")
        grid = tuple(params.get("grid", (1, 1, 1)))
        if not from_file:
            return self._accurate_correction(t, params["coords"], params["ensemble"], params.get("max_repetition", 20), grid)
        if self.results_folder is None:
            raise ValueError("results_folder was not given to CoordsToImageTransformer")
        prob_map = np.load(str(self.results_folder / "seg" / (params.get("format", "prob%06d.npy") % t)))
        return self._accurate_correction(prob_map, params["coords"], params["ensemble"], params.get("max_repetition", 20), grid), None

    def _accurate_correction(self, prob_map, coords: Coordinates, ensemble: bool, max_repetition: int = 20, grid=(1, 1, 1)):
        """prob_map: numpy / cuda tensor (x, y, z); `grid` repeats it like the reference (:432)."""
        import ctypes as C
        from . import _dev, _lib
        t = _dev.torch(); L = _lib.lib()
        if self._dev is None:
            self._upload()
        if not hasattr(prob_map, "is_cuda"):
            pm = np.asarray(prob_map)
            if tuple(grid) != (1, 1, 1):
                pm = np.repeat(np.repeat(np.repeat(pm, grid[1], axis=0), grid[2], axis=1), grid[0], axis=2)
            if pm.shape != self.proofed_shape:
                pm = pm[:self.proofed_shape[0], :self.proofed_shape[1], :self.proofed_shape[2]]
            prob_d = t.from_numpy(np.ascontiguousarray(pm, dtype=np.float32)).cuda()
        else:
            prob_d = prob_map.to(t.float32).contiguous()
        if tuple(prob_d.shape) != self.proofed_shape:
            raise ValueError(f"probability map shape {tuple(prob_d.shape)} != segmentation shape {self.proofed_shape}")
        n = len(self.subregions)
        boundary_ids = set(self.get_cells_on_boundary(coords.real, ensemble=ensemble).tolist())
        missed = np.zeros(n, dtype=np.uint8)
        for b in boundary_ids:
            missed[b - 1] = 1
        bbox, subs, offs, vol1 = self._dev
        if bbox.device != prob_d.device:
            self._dev = tuple(x.to(prob_d.device) for x in self._dev)
            bbox, subs, offs, vol1 = self._dev
        cur = t.from_numpy(np.ascontiguousarray(coords._raw, dtype=np.float32)).to(prob_d.device)
        missed_d = t.from_numpy(missed).to(prob_d.device)          # stays referenced until the call has been issued
        ws = _dev.workspace(L.ct_correction_workspace_bytes(_lib.ivec(self.proofed_shape), n), cur.device)
        iters = C.c_int(0)
        rc = L.ct_accurate_correction(prob_d.data_ptr(), _lib.ivec(self.proofed_shape), self.interpolation_factor, n, bbox.data_ptr(),
                                      subs.data_ptr(), offs.data_ptr(), missed_d.data_ptr(), vol1.data_ptr(),
                                      cur.data_ptr(), int(max_repetition), C.byref(iters), ws.data_ptr(), ws.numel(), _dev.stream(cur.device))
        if rc == -2:
            raise ValueError(f"Slices are out of range for image of size {self.proofed_shape}")
        _lib.check(rc, "ct_accurate_correction")
        self.last_iterations = iters.value
        return Coordinates(cur.cpu().numpy(), self.interpolation_factor, self.voxel_size, dtype="raw")
