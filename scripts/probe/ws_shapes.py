"""Device watershed against the oracle on a few more shapes (deep stack, wide thin stack, odd extents).  python scripts/probe/ws_shapes.py"""
import sys, time, importlib
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from oracle import watershed_ref as wr
from _ws_cases import random_case
seg = importlib.import_module('3deecelltracker_amd.segment')
for shape, n, zr, ms in (((256, 256, 64), 300, 2.0, 20), ((768, 640, 8), 500, 5.0, 10), ((333, 217, 37), 260, 3.0, 15), ((128, 128, 128), 200, 1.0, 25)):
    prob = random_case(shape, n, seed=sum(shape), specks=False)
    t0 = time.time(); want = wr.segment_centroids(prob, zr, "min_size", ms); t1 = time.time()
    got = seg.watershed_centroids(prob, zr, "min_size", ms, 0)
    ok = np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]) and (got[2], got[3]) == (want[2], want[3])
    print(shape, "cells", want[3], "oracle %.1f s" % (t1 - t0), "identical" if ok else f"DIFFERENT ({int((got[0] != want[0]).sum())} voxels)", flush=True)
