"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the marker-watershed region step of the legacy Tracker.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Reference: CellTracker/watershed.py:16-53 (watershed_2d), :55-108 (watershed_3d), called from tracker.py:671-684 (Tracker._watershed):
per z slice  EDT -> Gaussian(2) -> peak_local_max(min_distance=7) -> label -> watershed(-dist, markers, mask) -> find_boundaries(outer);
the boundaries are removed from the thresholded map, then in 3D  EDT(sampling 1, 1, z_xy_ratio) -> Gaussian(2, 2, 0.3) ->
peak_local_max(min_distance=3, exclude_border=0) -> label -> watershed -> min_size / cell_num -> remove_small_objects ->
relabel_sequential.

PARITY STATUS: **pinned** against the reference's own code running on the real scikit-image.  scipy IS installed here, so
distance_transform_edt, gaussian_filter, maximum_filter, label and center_of_mass are the very functions the reference runs.  scikit-image is
not importable by the image's main interpreter, so its four functions are RESTATED below -- but the image's second interpreter
(/opt/conda/bin/python3.9) carries scikit-image 0.18.3 (the generation `peak_local_max(indices=False)` needs): tests/golden/
make_watershed_golden.py runs CellTracker/watershed.py and Tracker._watershed there, unmodified, and records every stage
(tests/golden/watershed_skimage.npz); tests/test_watershed_pin.py holds each restatement to scikit-image's output exactly (fed with the
recorded output of the stage before it) and the composite to the reference's segmentation on eight volumes incl. the 512 x 512 x 32
benchmark stack (all 8.4 M voxels), the device path on the same.  The restatements:
  * peak_local_max  -- skimage/feature/peak.py (0.18): image == maximum_filter(image, footprint (2 d + 1)^ndim, mode='constant'),
    no peaks for a trivial (constant) image, & image > threshold with threshold = image.min() (no absolute / relative threshold given),
    border of width min_distance excluded unless exclude_border is 0/False, then the peaks in descending intensity with every peak
    CLOSER THAN min_distance (Chebyshev, strict: two peaks exactly min_distance apart both stay) to an already kept one dropped
    (ensure_spacing).  Inside the maximum filter's footprint two surviving peaks can only be that close if they are EQUAL, so the last
    step only acts on exact ties -- and WHICH of several exactly equal candidates stay is decided by the order np.argsort(-intensities)
    leaves them in: an unstable sort.  Every numpy before 1.25 (every numpy the reference's tensorflow==2.11 runs with) uses its generic
    introsort, restated here as argsort_quicksort and checked against numpy itself; the golden vectors are recorded on that code path, and
    the oracle equals them candidate for candidate.  (numpy >= 1.25 on a CPU with AVX-512 dispatches argsort to a network sort with another
    order among equal keys: there the reference itself keeps the other pixel of a tied pair -- 45 pairs / 96 voxels on the benchmark stack.)
  * watershed       -- skimage/segmentation/_watershed_cy.pyx: a priority queue of (value, age), seeded with all marker pixels (age 0),
    pop the smallest, give every unlabelled in-mask neighbour (connectivity 1, in ascending raveled-offset order) the popped pixel's
    label at PUSH time and push it with value = image[neighbour] and the next age.  All markers enter with age 0, so seeds of EXACTLY
    equal height are popped in the order upstream's binary heap happens to hold them (its array layout, global over the image): restated
    as _UpstreamHeap (seed_order="upstream": with it the oracle equals the reference on ALL nine recorded volumes, the designed tie volume
    included).  It is the DEFAULT, and since round 4 the device follows it too: mask components in which two seeds of exactly equal height
    meet make their group (z slice / volume) take a sequential replay of that heap on the device (ws_flood_upstream_kernel); everywhere else
    the order among equal seeds cannot matter and the component-parallel flood runs.  seed_order="raveled" (smaller raveled index first, the
    device's rule until round 3) is kept to show where the two differ: only where two equal seeds share a basin (the designed tie volume;
    not the benchmark stack, not the random volumes) -- the boundary between them moves by a row.
  * find_boundaries(mode='outer') -- skimage/segmentation/boundaries.py: grey dilation != grey erosion over the connectivity-c
    structure, kept where the pixel is background or the full-connectivity neighbourhood holds two different OBJECT labels.
  * remove_small_objects on a label image (sizes by bincount of the labels as they are) and relabel_sequential.
"""
from __future__ import annotations

import heapq

import numpy as np
import scipy.ndimage as ndi


# ------------------------------------------------------------------------------------------------ numpy's argsort, restated
def _msb(n: int) -> int:
    d = 0
    n >>= 1
    while n:
        d += 1
        n >>= 1
    return d


def _aheapsort(v, tosort, lo, n):
    """numpy/core/src/npysort/heapsort.cpp aheapsort_ on tosort[lo : lo + n] (1-based heap indices as upstream)"""
    def a(i): return tosort[lo + i - 1]
    def seta(i, x): tosort[lo + i - 1] = x
    for i in range(n >> 1, 0, -1):
        tmp = a(i); ii = i; j = ii << 1
        while j <= n:
            if j < n and v[a(j)] < v[a(j + 1)]:
                j += 1
            if v[tmp] < v[a(j)]:
                seta(ii, a(j)); ii = j; j += j
            else:
                break
        seta(ii, tmp)
    nn = n
    while nn > 1:
        tmp = a(nn); seta(nn, a(1)); nn -= 1
        i = 1; j = 2
        while j <= nn:
            if j < nn and v[a(j)] < v[a(j + 1)]:
                j += 1
            if v[tmp] < v[a(j)]:
                seta(i, a(j)); i = j; j += j
            else:
                break
        seta(i, tmp)


def argsort_quicksort(v) -> np.ndarray:
    """np.argsort(v) (kind='quicksort') of a float array without NaNs as numpy's generic, non-SIMD code computes it
    (numpy/core/src/npysort/quicksort.cpp aquicksort_: introsort -- median of three, Hoare partition with the pivot parked at pr - 1, the larger
    part pushed, ranges of <= 16 elements finished by insertion sort, heapsort for a popped range beyond depth 2 floor(log2 n)).  This is what every
    numpy before 1.25 runs, and later ones on CPUs without AVX-512 (there the dispatch goes to a network sort with another order among equal keys).
    Checked against numpy itself on 3012 arrays incl. heavy ties and the heapsort fallback (tests/test_watershed_pin.py runs the comparison wherever
    the image's second interpreter is present, with NPY_DISABLE_CPU_FEATURES)."""
    v = np.asarray(v, dtype=np.float64)
    num = len(v)
    tosort = list(range(num))
    if num < 2:
        return np.array(tosort, dtype=np.intp)
    pl, pr = 0, num - 1
    stack = []
    cdepth = _msb(num) * 2
    while True:
        if cdepth < 0:
            _aheapsort(v, tosort, pl, pr - pl + 1)
        else:
            while (pr - pl) > 15:
                pm = pl + ((pr - pl) >> 1)
                if v[tosort[pm]] < v[tosort[pl]]:
                    tosort[pm], tosort[pl] = tosort[pl], tosort[pm]
                if v[tosort[pr]] < v[tosort[pm]]:
                    tosort[pr], tosort[pm] = tosort[pm], tosort[pr]
                if v[tosort[pm]] < v[tosort[pl]]:
                    tosort[pm], tosort[pl] = tosort[pl], tosort[pm]
                vp = v[tosort[pm]]
                pi = pl; pj = pr - 1
                tosort[pm], tosort[pj] = tosort[pj], tosort[pm]
                while True:
                    pi += 1
                    while v[tosort[pi]] < vp:
                        pi += 1
                    pj -= 1
                    while vp < v[tosort[pj]]:
                        pj -= 1
                    if pi >= pj:
                        break
                    tosort[pi], tosort[pj] = tosort[pj], tosort[pi]
                pk = pr - 1
                tosort[pi], tosort[pk] = tosort[pk], tosort[pi]
                cdepth -= 1
                if pi - pl < pr - pi:
                    stack.append((pi + 1, pr, cdepth)); pr = pi - 1
                else:
                    stack.append((pl, pi - 1, cdepth)); pl = pi + 1
            for pi in range(pl + 1, pr + 1):
                vi = tosort[pi]; vp = v[vi]; pj = pi; pk = pi - 1
                while pj > pl and vp < v[tosort[pk]]:
                    tosort[pj] = tosort[pk]; pj -= 1; pk -= 1
                tosort[pj] = vi
        if not stack:
            break
        pl, pr, cdepth = stack.pop()
    return np.array(tosort, dtype=np.intp)


# ------------------------------------------------------------------------------------------------ restated skimage functions
def peak_local_max_mask(image: np.ndarray, min_distance: int, exclude_border=True) -> np.ndarray:
    """skimage.feature.peak_local_max(image, min_distance=..., exclude_border=..., indices=False) -> bool mask."""
    image = np.asarray(image, dtype=np.float64)
    size = 2 * int(min_distance) + 1
    image_max = ndi.maximum_filter(image, footprint=np.ones((size,) * image.ndim, dtype=bool), mode="constant")
    out = image == image_max
    if np.all(out):                                   # "no peak for a trivial image"
        out[:] = False
    out &= image > image.min()
    if exclude_border is True:
        border = int(min_distance)
    else:
        border = int(exclude_border)                  # 0 / False: nothing excluded
    if border > 0:
        for ax in range(image.ndim):
            sl = [slice(None)] * image.ndim
            sl[ax] = slice(None, border); out[tuple(sl)] = False
            sl[ax] = slice(-border, None); out[tuple(sl)] = False
    # ensure_spacing: descending intensity (ties: raveled index ascending), drop what is within min_distance of a kept peak
    idx = np.flatnonzero(out)
    if idx.size > 1:
        vals = image.ravel()[idx]
        order = argsort_quicksort(-vals)              # upstream: np.argsort(-intensities) over the candidates in raveled order
        coords = np.stack(np.unravel_index(idx[order], image.shape), axis=1)
        kept = []
        keep_mask = np.zeros(len(coords), dtype=bool)
        for i, c in enumerate(coords):
            if kept and np.any(np.max(np.abs(np.asarray(kept) - c), axis=1) < min_distance):     # (strict: peaks exactly min_distance apart both stay)
                continue
            kept.append(c); keep_mask[i] = True
        out[:] = False
        out.ravel()[idx[order][keep_mask]] = True
    return out


def label_full(mask: np.ndarray) -> np.ndarray:
    """skimage.morphology.label(mask) (connectivity = ndim): raster-order numbering, like scipy's."""
    lab, _ = ndi.label(mask, structure=np.ones((3,) * mask.ndim, dtype=bool))
    return lab.astype(np.int32)


class _UpstreamHeap:
    """skimage/segmentation/heap_general.pxi + heap_watershed.pxi restated: a textbook binary heap over (value, age) -- push appends and sifts
    up while STRICTLY smaller than the parent; pop moves the last element to the root and sifts it down towards the strictly smaller child
    (the left one when both children tie).  Elements that compare equal -- only the seeds can: every later element has its own age -- come out
    in an order that depends on this array layout.  Checked against skimage.segmentation.watershed itself on 60 tie-heavy inputs."""

    def __init__(self):
        self.h = []

    @staticmethod
    def _smaller(a, b):
        return a[0] < b[0] or (a[0] == b[0] and a[1] < b[1])

    def push(self, e):
        h = self.h
        h.append(e)
        c = len(h) - 1
        while c > 0:
            p = (c + 1) // 2 - 1
            if self._smaller(h[c], h[p]):
                h[c], h[p] = h[p], h[c]; c = p
            else:
                break

    def pop(self):
        h = self.h
        top = h[0]
        last = h.pop()
        if h:
            h[0] = last
            i, n = 0, len(h)
            while True:
                l, r, sm = 2 * i + 1, 2 * i + 2, i
                if l >= n:
                    break
                if self._smaller(h[l], h[i]):
                    sm = l
                if r < n and self._smaller(h[r], h[sm]):
                    sm = r
                if sm == i:
                    break
                h[i], h[sm] = h[sm], h[i]; i = sm
        return top

    def __len__(self):
        return len(self.h)


def watershed(image: np.ndarray, markers: np.ndarray, mask: np.ndarray, seed_order: str = "upstream") -> np.ndarray:
    """skimage.segmentation.watershed(image, markers, mask=mask) (connectivity 1, no compactness, no watershed line).
    seed_order: how seeds of EXACTLY equal height are popped -- "upstream" (default: the order scikit-image's own heap leaves them in,
    _UpstreamHeap; the device path follows it since round 4) or "raveled" (smaller raveled index first: the device's rule until round 3, kept to
    show where the two differ).  Every other element carries its own age, so the two only differ where two equal seeds share a basin."""
    image = np.asarray(image, dtype=np.float64)
    shape = image.shape
    out = np.where(mask, markers, 0).astype(np.int32).ravel().copy()      # markers outside the mask are dropped (skimage: markers[~mask] = 0)
    img = image.ravel()
    msk = np.asarray(mask, dtype=bool).ravel()
    strides = [int(np.prod(shape[a + 1:])) for a in range(len(shape))]
    offs = sorted([(-s, a, -1) for a, s in enumerate(strides)] + [(s, a, 1) for a, s in enumerate(strides)])   # ascending raveled offset
    coords_of = lambda i: np.unravel_index(i, shape)
    seeds = np.flatnonzero(out)
    if seed_order == "upstream":
        # Equal seeds only matter where they share a connectivity-1 component of the mask (components never interact, and inside one the
        # pops of the global sequence that belong to it are ordered by its own (value, age) keys -- seeds apart, all distinct).  Without such a
        # pair the result does not depend on the order among equal seeds and the C heapq path computes it (the same criterion sends a
        # group to the sequential replay on the device; the pin tests hold both to scikit-image's output on every recorded volume).
        comp = ndi.label(msk.reshape(shape))[0].ravel()[seeds]
        if np.unique(np.stack([comp.astype(np.float64), img[seeds]], 1), axis=0).shape[0] == seeds.size:
            seed_order = "raveled"
    if seed_order == "upstream":
        up = _UpstreamHeap()
        for i in np.flatnonzero(out):
            up.push((img[i], 0, int(i)))
        push, pop, pending = up.push, up.pop, up
        age = 1
    elif seed_order == "raveled":
        heap = [(img[i], 0, int(i)) for i in np.flatnonzero(out)]
        heapq.heapify(heap)
        push, pop, pending = (lambda e: heapq.heappush(heap, e)), (lambda: heapq.heappop(heap)), heap
        age = 0
    else:
        raise ValueError("seed_order is 'raveled' or 'upstream'")
    while len(pending):
        _, _, i = pop()
        ci = coords_of(i)
        for off, ax, sgn in offs:
            c = ci[ax] + sgn
            if c < 0 or c >= shape[ax]:
                continue
            j = i + off
            if not msk[j] or out[j]:
                continue
            age += 1
            out[j] = out[i]
            push((img[j], age, int(j)))
    return out.reshape(shape)


def find_boundaries_outer(labels: np.ndarray, connectivity: int) -> np.ndarray:
    """skimage.segmentation.find_boundaries(labels, connectivity=c, mode='outer', background=0)."""
    lab = np.asarray(labels)
    nd = lab.ndim
    fp = ndi.generate_binary_structure(nd, connectivity)
    boundaries = ndi.grey_dilation(lab, footprint=fp) != ndi.grey_erosion(lab, footprint=fp)      # (scipy's default border: reflect)
    bg = lab == 0
    full = ndi.generate_binary_structure(nd, nd)
    inv = lab.copy()
    inv[bg] = np.iinfo(lab.dtype).max
    adjacent = (ndi.grey_dilation(lab, footprint=full) != ndi.grey_erosion(inv, footprint=full)) & ~bg
    return boundaries & (bg | adjacent)


def remove_small_objects(labels: np.ndarray, min_size: int) -> np.ndarray:
    sizes = np.bincount(labels.ravel())
    small = sizes < min_size
    out = labels.copy()
    out[small[labels]] = 0
    return out


def relabel_sequential(labels: np.ndarray) -> np.ndarray:
    present = np.unique(labels)
    present = present[present > 0]
    lut = np.zeros(int(labels.max()) + 1, dtype=np.int32)
    lut[present] = np.arange(1, present.size + 1, dtype=np.int32)
    return lut[labels]


# ------------------------------------------------------------------------------------------------ the reference's functions
def watershed_2d(image_pred: np.ndarray, z_range: int, min_distance: int = 7, collect=None, peaks=None, seed_order: str = "upstream"):
    """watershed.py:16-53 -> (bn_output, boundary).  peaks (tests only): a recorded peak mask [x, y, z] used instead of peak_local_max's --
    the choice among exactly tied candidates that one particular upstream run made (see the header)."""
    image_pred = np.asarray(image_pred)
    boundary = np.zeros(image_pred.shape, dtype=bool)
    for z in range(z_range):
        bn = image_pred[:, :, z] > 0.5
        dist = ndi.distance_transform_edt(bn, sampling=[1, 1])
        dist_smooth = ndi.gaussian_filter(dist, 2, mode="constant")
        local_maxi = peak_local_max_mask(dist_smooth, min_distance=min_distance) if peaks is None else np.asarray(peaks[:, :, z], dtype=bool)
        markers = label_full(local_maxi)
        labels_ws = watershed(-dist_smooth, markers, bn, seed_order)
        boundary[:, :, z] = find_boundaries_outer(labels_ws, connectivity=2)
        if collect is not None:
            collect.append({"dist": dist, "dist_smooth": dist_smooth, "peaks": local_maxi, "labels": labels_ws})
    bn_output = image_pred > 0.5
    bn_output[boundary] = False
    return bn_output, boundary


def watershed_3d(image_watershed2d: np.ndarray, samplingrate, method: str, min_size: int, cell_num: int, min_distance: int, collect=None, peaks=None,
                 seed_order: str = "upstream"):
    """watershed.py:55-108 -> (labels_wo_bd, labels_clear, min_size, cell_num).  peaks: as in watershed_2d."""
    dist = ndi.distance_transform_edt(image_watershed2d, sampling=samplingrate)
    dist_smooth = ndi.gaussian_filter(dist, (2, 2, 0.3), mode="constant")
    local_maxi = peak_local_max_mask(dist_smooth, min_distance=min_distance, exclude_border=0) if peaks is None else np.asarray(peaks, dtype=bool)
    markers = label_full(local_maxi)
    labels_ws = watershed(-dist_smooth, markers, image_watershed2d, seed_order)
    counts = np.sort(np.bincount(labels_ws.ravel()))
    if method == "min_size":
        cell_num = int(np.sum(counts >= min_size) - 1)
    elif method == "cell_num":
        min_size = int(counts[-cell_num - 1])
    else:
        raise ValueError("The method parameter should be either min_size or cell_num")
    labels_clear = remove_small_objects(labels_ws, min_size)
    labels_bd = find_boundaries_outer(labels_clear, connectivity=3)
    labels_wo_bd = labels_clear.copy()
    labels_wo_bd[labels_bd] = 0
    labels_wo_bd = remove_small_objects(labels_wo_bd, min_size)
    if collect is not None:
        collect.append({"dist": dist, "dist_smooth": dist_smooth, "peaks": local_maxi, "labels": labels_ws})
    return labels_wo_bd, labels_clear, min_size, cell_num


def tracker_watershed(image_cell_bg_xyz: np.ndarray, z_xy_ratio: float, method: str = "min_size", min_size: int = 0, cell_num: int = 0,
                      peaks2d=None, peaks3d=None, seed_order: str = "upstream"):
    """Tracker._watershed (tracker.py:671-684) -> (segmentation_auto int32, min_size, cell_num)."""
    img = np.asarray(image_cell_bg_xyz)
    wo_border, _ = watershed_2d(img, z_range=img.shape[2], min_distance=7, peaks=peaks2d, seed_order=seed_order)
    _, wi_border, min_size, cell_num = watershed_3d(wo_border, [1, 1, z_xy_ratio], method, min_size, cell_num, min_distance=3, peaks=peaks3d,
                                                    seed_order=seed_order)
    return relabel_sequential(wi_border), min_size, cell_num


def segment_centroids(image_cell_bg_xyz: np.ndarray, z_xy_ratio: float, method: str = "min_size", min_size: int = 0, cell_num: int = 0,
                      seed_order: str = "upstream"):
    """-> (labels int32, centres float64 [n, 3] via the reference's center_of_mass call (tracker.py:646-647), min_size, cell_num)."""
    labels, min_size, cell_num = tracker_watershed(image_cell_bg_xyz, z_xy_ratio, method, min_size, cell_num, seed_order=seed_order)
    n = int(labels.max())
    centres = np.asarray(ndi.center_of_mass(labels > 0, labels, range(1, n + 1)), dtype=np.float64).reshape(n, 3) if n else np.zeros((0, 3))
    return labels, centres, min_size, cell_num
