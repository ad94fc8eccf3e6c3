"""How sparse is the marker watershed's work on the benchmark's kind of volume?  (round-5 verdict, item 2: "run the passes over listed tiles only".)
For a 512 x 512 x 32 probability map with 600 cells: the fraction of voxels / (x, y) columns / 8 x 8 x z tiles at which each intermediate of
watershed.py:16-108 is NOT the background constant, i.e. what an active-tile scheme would still have to process.  CPU only (numpy / scipy)."""
import importlib
import sys
import os
import numpy as np
from scipy import ndimage as ndi
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
synth = importlib.import_module("3deecelltracker_amd.synth")


def frac(mask, tile=None):
    if tile is None:
        return float(mask.mean())
    X, Y, Z = mask.shape
    tx, ty = tile
    m = mask.any(axis=2)
    m = m[:X - X % tx, :Y - Y % ty].reshape(X // tx, tx, Y // ty, ty).any(axis=(1, 3))
    return float(m.mean())


def dilate_xy(mask, r):
    st = np.ones((2 * r + 1, 2 * r + 1, 1), dtype=bool)
    return ndi.binary_dilation(mask, structure=st)


for name, prob in (("make_prob_map(600 cells, radius 4,4,1.5)", synth.make_prob_map(0, (512, 512, 32), 600)),
                   ("make_prob_map(600 cells, radius 6,6,2)", synth.make_prob_map(0, (512, 512, 32), 600, radius=(6.0, 6.0, 2.0)))):
    fg = prob > 0.5
    print(f"== {name}: foreground {100 * frac(fg):.1f} % of the voxels, {100 * frac(fg.any(axis=2)[..., None]):.1f} % of the (x, y) columns")
    # 2-D stage, per z slice: EDT support = fg; Gaussian radius 8 -> fg (+) 8; window maximum r = 7 reads fg (+) 15 and is non-zero there
    for r, what in ((0, "mask / EDT"), (8, "smoothed EDT (Gaussian radius 8)"), (15, "window maximum (min_distance 7) non-zero"), (22, "what the maximum pass READS around its non-zero outputs")):
        d = dilate_xy(fg, r) if r else fg
        print(f"   2-D stage, {what:58s}: {100 * frac(d):5.1f} % of voxels   {100 * frac(d, (8, 8)):5.1f} % of 8 x 8 x z tiles   {100 * frac(d.any(axis=2)[..., None]):5.1f} % of columns")
    # 3-D stage: the Gaussian and the maximum also run along z (radius_z up to 7 of 32 slices): every column that is active in xy is active in all z
    d3 = np.broadcast_to(dilate_xy(fg, 8).any(axis=2)[..., None], fg.shape)
    print(f"   3-D stage, smoothed EDT (xy radius 8, z radius covering the stack) : {100 * frac(d3):5.1f} % of voxels")
