"""ORACLE (test infrastructure -- never imported by the product path).

numpy restatement of the reference's accurate correction of cell centres:

* move_cells           <- reference CellTracker/coord_image_transformer.py:292-369
* _correction_once     <- reference CellTracker/coord_image_transformer.py:449-489
* accurate_correction  <- reference CellTracker/coord_image_transformer.py:406-447 (the coordinate loop; the final
                           label image goes through skimage's watershed -- absent here and out of scope)
* get_cells_on_boundary <- reference CellTracker/coord_image_transformer.py:371-404

PARITY STATUS: PINNED against golden vectors produced by the reference's own accurate_correction on hand-built
sub-regions (tests/golden/make_golden.py -> tests/golden/correction.npz).  Coordinates are float32 "raw" voxel
coordinates exactly like the reference's `Coordinates` type.
"""
from __future__ import annotations

import numpy as np


def raw_to_interp(raw_f32: np.ndarray, factor: int) -> np.ndarray:
    """Coordinates.interp (coord_image_transformer.py:110-122)."""
    return np.round(raw_f32 * np.asarray((1, 1, factor))[None, :]).astype(np.int32)


def move_cells(shape_xyz, factor, subregions, movements_nx3, cells_missed):
    """-> (label sums, overlap counts) on the z-interpolated grid."""
    sx, sy, sz = shape_xyz
    out = np.zeros((sx, sy, sz * factor), dtype=np.int64)
    mask = np.zeros_like(out)
    dims = (sx, sy, sz * factor)
    for i, (bbox, sub) in enumerate(subregions):
        if (i + 1) in cells_missed:
            continue
        dst, src = [], []
        for s, c, size in zip(bbox, movements_nx3[i], dims):
            a_ = s.start + int(c); a = max(a_, 0)
            b_ = s.stop + int(c); b = min(b_, size)
            if a >= b:
                raise ValueError(f"Slices are out of range for image of size {dims}")
            dst.append(slice(a, b)); src.append(slice(a - a_, (s.stop - s.start) - (b_ - b)))
        part = sub[tuple(src)]
        out[tuple(dst)] += part.astype(np.int64) * (i + 1)
        mask[tuple(dst)] += part.astype(np.int64)
    return out, mask


def correction_once(prob, shape_xyz, factor, z_slice, subregions, n_labels, vol1_raw, coords_raw, cells_missed):
    mov = raw_to_interp(coords_raw - vol1_raw, factor)
    labels_i, mask_i = move_cells(shape_xyz, factor, subregions, mov, cells_missed)
    labels = labels_i[:, :, z_slice].copy(); mask = mask_i[:, :, z_slice]
    labels[mask > 1] = 0
    pos = np.full((n_labels, 3), np.nan)
    gx, gy, gz = np.meshgrid(*(np.arange(s, dtype=np.float64) for s in labels.shape), indexing="ij")
    p64 = prob.astype(np.float64)
    for lab in range(1, n_labels + 1):
        sel = labels == lab
        w = p64[sel].sum()
        if sel.any() and w != 0:
            pos[lab - 1] = ((p64[sel] * gx[sel]).sum() / w, (p64[sel] * gy[sel]).sum() / w, (p64[sel] * gz[sel]).sum() / w)
    lost = np.isnan(pos[:, 0])
    pos[lost] = np.round(coords_raw).astype(np.int32)[lost]
    new_raw = pos.astype(np.float32)
    return new_raw, new_raw - coords_raw


def accurate_correction(prob, shape_xyz, factor, subregions, n_labels, vol1_raw, coords_raw, cells_missed, max_repetition=20):
    z_slice = slice(factor // 2, factor * shape_xyz[2], factor)
    cur = np.asarray(coords_raw, dtype=np.float32)
    it = 0
    for it in range(1, max_repetition + 1):
        cur, delta = correction_once(prob, shape_xyz, factor, z_slice, subregions, n_labels, np.asarray(vol1_raw, np.float32), cur, cells_missed)
        if np.max(raw_to_interp(delta, factor)) < 0.5:
            break
    return cur, it


def get_cells_on_boundary(coords_real, shape_xyz, voxel_size, ensemble, boundary_xy=6):
    if ensemble:
        boundary_xy = 0
    x, y, z = coords_real.T
    sx, sy, sz = shape_xyz
    near = ((x < boundary_xy) | (y < boundary_xy) | (x > (sx - boundary_xy) * voxel_size[0]) |
            (y > (sy - boundary_xy) * voxel_size[1]) | (z < 0) | (z > sz * voxel_size[2]))
    return np.where(near)[0] + 1
