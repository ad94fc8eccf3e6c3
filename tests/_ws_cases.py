"""Synthetic probability maps of the watershed tests (numpy only: also imported by tests/golden/make_watershed_golden.py, which runs under
another interpreter).  Every builder is deterministic; the golden file records the sha256 of each map it was made from."""
import hashlib

import numpy as np


def sha(a) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def blobs(shape, centres, radii, z_flat=3.0, level=0.9):
    g = np.stack(np.meshgrid(*(np.arange(s) for s in shape), indexing="ij"), -1).astype(float)
    prob = np.zeros(shape, np.float32)
    for c, r in zip(centres, radii):
        prob[(((g - np.asarray(c, float)) / np.array([r, r, r / z_flat])) ** 2).sum(-1) <= 1.0] = level
    return prob


def blobs_local(shape, centres, radii, z_flat=3.0, level=0.9):
    """blobs() painted window by window (the same voxels: an ellipsoid's mask only needs its bounding box) -- for stacks of 10^7 voxels and
    thousands of cells, where blobs() evaluates every ellipsoid on the whole grid."""
    prob = np.zeros(shape, np.float32)
    for c, r in zip(centres, radii):
        c = np.asarray(c, float); rad = np.array([r, r, r / z_flat])
        lo = np.maximum(np.floor(c - rad).astype(int), 0); hi = np.minimum(np.ceil(c + rad).astype(int) + 1, shape)
        g = np.stack(np.meshgrid(*(np.arange(lo[a], hi[a]) for a in range(3)), indexing="ij"), -1).astype(float)
        sub = prob[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]]
        sub[(((g - c) / rad) ** 2).sum(-1) <= 1.0] = level
    return prob


def random_case_large(shape, n, seed):
    """random_case(..., specks=False) for large stacks (blobs_local; same recipe, its own random stream for the plateau noise)"""
    rng = np.random.default_rng(seed)
    lo = np.array([8, 8, 2]); hi = np.array([shape[0] - 8, shape[1] - 8, shape[2] - 2])
    c = rng.uniform(lo, hi, (n, 3))
    prob = blobs_local(shape, c, rng.uniform(4, 8, n), level=0.8)
    prob += rng.uniform(0, 0.2, shape).astype(np.float32) * (prob > 0)       # ragged plateau
    return prob


def touching_case():
    """two overlapping blobs (one connected component), one isolated blob, one blob below min_size"""
    return blobs((96, 96, 12), [(30, 30, 6), (45, 30, 6), (70, 70, 5), (20, 75, 3)], [9, 9, 8, 2.2])


def tie_case():
    """Shapes whose EDT has exact ties: a long bar (a ridge of EQUAL smoothed-EDT maxima: ensure_spacing has to thin it in raveled
    order), two identical touching squares (two markers of equal value in ONE mask component: the flood's tie rule decides the boundary),
    a symmetric cross, the same bar again in other slices and a thick slab spanning several z (ties in the 3-D stage too)."""
    prob = np.zeros((96, 80, 10), np.float32)
    prob[10:70, 8:17, 1:4] = 0.9                       # bar, 60 x 9, three slices
    prob[20:33, 30:43, 2] = 0.9; prob[33:46, 30:43, 2] = 0.9      # two 13 x 13 squares sharing an edge
    prob[60:81, 50:53, 5:8] = 0.9; prob[69:72, 41:62, 5:8] = 0.9  # cross
    prob[12:40, 56:70, 6:9] = 0.9                      # slab
    return prob


def wide_front_case():
    """Two large overlapping discs per slice (one component of ~4500 pixels inside a 100 x 60 box, two markers): the flood's frontier grows to a
    few hundred entries, i.e. rounds of the batched LDS flood with more members than lanes; a ragged plateau keeps the distances irregular."""
    prob = blobs((128, 96, 4), [(42, 48, 1.5), (84, 48, 1.5)], [29, 27], z_flat=1.0, level=0.8)
    prob += np.random.default_rng(3).uniform(0, 0.2, prob.shape).astype(np.float32) * (prob > 0)
    return prob


def random_case(shape, n, seed, specks=True):
    rng = np.random.default_rng(seed)
    lo = np.array([8, 8, 2]); hi = np.array([shape[0] - 8, shape[1] - 8, shape[2] - 2])
    c = rng.uniform(lo, hi, (n, 3))
    prob = blobs(shape, c, rng.uniform(4, 8, n), level=0.8)
    prob += rng.uniform(0, 0.2, shape).astype(np.float32) * (prob > 0)       # ragged plateau
    if specks:                                                                # isolated specks (dropped by min_size): identical shapes,
        prob[rng.uniform(size=shape) > 0.995] = 0.7                           # i.e. EXACT ties of their smoothed-EDT maxima
    return prob


def headline_case(make_stack):
    """512 x 512 x 32 / ~600 cells: the benchmark's stack (3deecelltracker_amd.synth.make_stack) as a probability map"""
    stack, _ = make_stack((512, 512, 32), 600, seed=0)
    return np.clip((stack.astype(np.float32) - 100.0) / 600.0, 0, 1)


# name -> (builder taking make_stack, z_xy_ratio, min_size).  Exact ties between peak candidates closer than min_distance are resolved upstream by
# an UNSTABLE argsort (skimage/feature/peak.py: np.argsort(-intensities); numpy >= 1.25 dispatches it to an AVX-512 network sort where the CPU
# has one), i.e. machine-dependently; the oracle fixes one admissible order.  TIE_FREE cases are compared stage by stage, the others (identical
# specks whose regions min_size removes anyway) on the final segmentation, "ties" only on what is order-independent.
PIN_CASES = {
    "touching": (lambda ms: touching_case(), 3.0, 40),
    "clean_a": (lambda ms: random_case((90, 70, 14), 25, 21, specks=False), 4.0, 15),
    "clean_b": (lambda ms: random_case((64, 80, 9), 14, 22, specks=False), 2.5, 10),
    "clean_c": (lambda ms: random_case((120, 100, 16), 60, 23, specks=False), 3.0, 20),
    "random_a": (lambda ms: random_case((90, 70, 14), 25, 11), 4.0, 15),
    "random_b": (lambda ms: random_case((64, 80, 9), 14, 12), 2.5, 10),
    "random_c": (lambda ms: random_case((120, 100, 16), 60, 13), 3.0, 20),
    "ties": (lambda ms: tie_case(), 2.0, 30),
    "headline": (headline_case, 4.0, 20),
}
TIE_FREE = ("touching", "clean_a", "clean_b", "clean_c")
FINAL_ONLY = ("random_a", "random_b", "random_c", "headline")
