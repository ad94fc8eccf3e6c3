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
