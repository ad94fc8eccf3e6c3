"""Seeded synthetic weights, image stacks and point sets.

There are no trained weights or images in the reference tree (they live on OSF), and no
network here, so parity tests and the benchmark run on seeded random-init weights of the
reference architectures and on synthetic stacks / point sets:

* U-Net / FFN weights: Glorot-uniform kernels, small normal biases, BatchNorm statistics drawn
  away from the identity so that a wrong epilogue order is visible.
* Point-set pairs follow the reference's own training-pair recipe: centred affine
  ``I + (U(0,1)-0.5)*0.2`` plus jitter (reference ffn.py:18,23-24,51-53) with a fraction of
  points replaced (segmentation errors, reference synthesize.py:52-72) and a random permutation.
* Stacks: Gaussian background + anisotropic Gaussian blobs, then a simple normalisation.

Weight container (plain dict of numpy arrays, Keras layouts):
    unet: {"arch": name, "convs": [ {"kernel": (3,3,3,Cin,Cout), "bias": (Cout,), "gamma", "beta",
           "mean", "var": (Cout,)} ... ], "head": {"kernel": (1,1,1,C,1), "bias": (1,)} }
    ffn:  {"w1": (61,512), "bn1": {gamma,beta,mean,var}, "w2": (1024,512), "bn2": {...},
           "w3": (512,1), "b3": (1,)}
"""
from __future__ import annotations

from pathlib import Path

import numpy as np

from .arch import ARCHS, FFN_FEAT, FFN_HID


def _glorot(rng, shape, fan_in, fan_out):
    lim = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=shape).astype(np.float32)


def _bn(rng, c):
    return {
        "gamma": rng.uniform(0.5, 1.5, c).astype(np.float32),
        "beta": rng.normal(0.0, 0.1, c).astype(np.float32),
        "mean": rng.normal(0.0, 0.1, c).astype(np.float32),
        "var": rng.uniform(0.5, 1.5, c).astype(np.float32),
    }


def make_unet_weights(arch_name: str = "unet3_a", seed: int = 0) -> dict:
    arch = ARCHS[arch_name]
    rng = np.random.default_rng(seed)
    convs = []
    for cin, cout in arch.conv_layers():
        layer = {
            "kernel": _glorot(rng, (3, 3, 3, cin, cout), 27 * cin, 27 * cout),
            "bias": rng.normal(0.0, 0.05, cout).astype(np.float32),
        }
        layer.update(_bn(rng, cout))
        convs.append(layer)
    c = arch.out[1]
    head = {"kernel": _glorot(rng, (1, 1, 1, c, 1), c, 1), "bias": rng.normal(0.0, 0.05, 1).astype(np.float32)}
    return {"arch": arch_name, "convs": convs, "head": head}


def make_ffn_weights(seed: int = 0, gain: float = 1.0, shift: float = 0.0) -> dict:
    """`gain` scales the last layer so that the sigmoid scores spread over (0,1) instead of
    clustering at 0.5 and `shift` moves the output bias (a trained FFN is strongly bimodal with
    most pairs near 0; a Glorot-init one is not)."""
    rng = np.random.default_rng(seed + 1000)
    return {
        "w1": _glorot(rng, (FFN_FEAT, FFN_HID), FFN_FEAT, FFN_HID),
        "bn1": _bn(rng, FFN_HID),
        "w2": _glorot(rng, (2 * FFN_HID, FFN_HID), 2 * FFN_HID, FFN_HID),
        "bn2": _bn(rng, FFN_HID),
        "w3": (_glorot(rng, (FFN_HID, 1), FFN_HID, 1) * gain).astype(np.float32),
        "b3": (rng.normal(0.0, 0.05, 1) + shift).astype(np.float32),
    }


def make_point_pair(n: int, seed: int = 0, box=(512.0, 512.0, 32.0), voxel_size=(1.0, 1.0, 4.0),
                    affine_level: float = 0.2, move_level: float = 0.001, replace_ratio: float = 0.15):
    """Return (X, Y): X uniform in the stack box (real units), Y = affine-perturbed, partly
    replaced, permuted copy (the reference's synthetic training-pair recipe, ffn.py:18-53)."""
    rng = np.random.default_rng(seed)
    ext = np.asarray(box, dtype=np.float64) * np.asarray(voxel_size, dtype=np.float64)
    x = rng.uniform(0.0, 1.0, (n, 3)) * ext[None, :]
    mean = x.mean(axis=0)
    scale = 3.0 * x.std()
    xc = (x - mean) / scale
    a = np.eye(3) + (rng.uniform(0, 1, (3, 3)) - 0.5) * affine_level
    y = xc @ a + (rng.uniform(0, 1, xc.shape) - 0.5) * 4 * move_level
    n_rep = int(round(n * replace_ratio))
    if n_rep:
        idx = rng.choice(n, n_rep, replace=False)
        y[idx] = (rng.uniform(0.0, 1.0, (n_rep, 3)) * ext[None, :] - mean) / scale
    perm = rng.permutation(n)
    y = y[perm]
    return x, y * scale + mean


def make_stack(shape=(512, 512, 32), n_cells: int = 600, seed: int = 0):
    """Synthetic (x, y, z) uint16 stack and its blob centres (voxel coords)."""
    rng = np.random.default_rng(seed)
    sx, sy, sz = shape
    img = rng.normal(100.0, 20.0, shape).astype(np.float32)
    np.clip(img, 0, None, out=img)
    lo = np.array([6, 6, 1]); hi = np.array([sx - 6, sy - 6, max(sz - 1, 2)])
    centres = rng.uniform(lo, hi, (n_cells, 3))
    amps = rng.uniform(400, 2000, n_cells)
    sig = np.array([3.0, 3.0, 1.0])
    rad = np.array([9, 9, 3])
    for c, a in zip(centres, amps):
        c0 = np.maximum(np.floor(c - rad).astype(int), 0)
        c1 = np.minimum(np.ceil(c + rad).astype(int) + 1, shape)
        gx = np.exp(-0.5 * ((np.arange(c0[0], c1[0]) - c[0]) / sig[0]) ** 2)
        gy = np.exp(-0.5 * ((np.arange(c0[1], c1[1]) - c[1]) / sig[1]) ** 2)
        gz = np.exp(-0.5 * ((np.arange(c0[2], c1[2]) - c[2]) / sig[2]) ** 2)
        img[c0[0]:c1[0], c0[1]:c1[1], c0[2]:c1[2]] += a * gx[:, None, None] * gy[None, :, None] * gz[None, None, :]
    return np.clip(img, 0, 65535).astype(np.uint16), centres


def normalize_stack(img_u16: np.ndarray) -> np.ndarray:
    """Cheap stand-in for the reference's pre-processing (median subtract, clamp, scale): the
    LCN step is SURVEY 8(f) 'next' row #1 and not on this path.  Returns fp32 (1,x,y,z,1)."""
    img = img_u16.astype(np.float32)
    img -= np.median(img)
    np.clip(img, 0, None, out=img)
    img /= (img.std() + 1e-6)
    return img[None, :, :, :, None]


def make_correction_case(seed=0, shape=(64, 56, 8), factor=5, n_cells=18, margin=8):
    """Hand-built state for the accurate correction: ellipsoidal sub-regions on the z-interpolated grid, a
    probability map whose blobs sit a few voxels away from the current positions (the reference builds its sub-regions with
    skimage, which is not available; the correction itself only needs (bbox, mask) pairs)."""
    rng = np.random.default_rng(seed)
    sx, sy, sz = shape
    cen = np.stack([rng.uniform(margin, sx - margin, n_cells), rng.uniform(margin, sy - margin, n_cells), rng.uniform(1.5, sz - 1.5, n_cells)], 1)
    subregions = []
    for c in cen:
        r = np.array([rng.integers(3, 6), rng.integers(3, 6), rng.integers(4, 9)])          # radii on the interp grid
        ci = np.array([c[0], c[1], c[2] * factor + factor // 2])
        lo = np.maximum(np.floor(ci - r).astype(int), 0); hi = np.minimum(np.ceil(ci + r).astype(int) + 1, (sx, sy, sz * factor))
        g = np.meshgrid(*(np.arange(lo[a], hi[a]) for a in range(3)), indexing="ij")
        sub = sum(((g[a] - ci[a]) / r[a]) ** 2 for a in range(3)) <= 1.0
        subregions.append((tuple(slice(int(lo[a]), int(hi[a])) for a in range(3)), sub))
    vol1 = cen.astype(np.float32)
    shift = np.stack([rng.uniform(-3, 3, n_cells), rng.uniform(-3, 3, n_cells), rng.uniform(-0.6, 0.6, n_cells)], 1)
    gx, gy, gz = np.meshgrid(np.arange(sx), np.arange(sy), np.arange(sz), indexing="ij")
    prob = np.zeros(shape, dtype=np.float64)
    for c in cen + shift:
        prob += np.exp(-0.5 * (((gx - c[0]) / 2.5) ** 2 + ((gy - c[1]) / 2.5) ** 2 + ((gz - c[2]) / 0.8) ** 2))
    prob = np.clip(prob, 0, 1).astype(np.float32)
    coords0 = (cen + 0.35 * shift + rng.normal(0, 0.2, cen.shape)).astype(np.float32)
    return {"shape": shape, "factor": factor, "subregions": subregions, "vol1": vol1, "prob": prob, "coords0": coords0,
            "voxel_size": np.array([1.0, 1.0, 4.0])}




def make_prob_map(seed: int, shape, n_cells: int, radius=(4.0, 4.0, 1.5), speckle: float = 0.0005):
    """Synthetic U-Net output: `n_cells` soft ellipsoids (peak ~0.95, many of them touching) plus isolated bright
    speckle voxels (regions that min_size must remove).  float32 [x, y, z] in [0, 1]."""
    rng = np.random.default_rng(seed)
    X, Y, Z = (int(s) for s in shape)
    prob = np.zeros((X, Y, Z), dtype=np.float32)
    rx, ry, rz = radius
    wx, wy, wz = int(3 * rx) + 1, int(3 * ry) + 1, int(3 * rz) + 1
    centres = rng.uniform(0, 1, size=(n_cells, 3)) * np.array([X - 1, Y - 1, Z - 1])
    for cx, cy, cz in centres:
        x0, x1 = max(0, int(cx) - wx), min(X, int(cx) + wx + 1)
        y0, y1 = max(0, int(cy) - wy), min(Y, int(cy) + wy + 1)
        z0, z1 = max(0, int(cz) - wz), min(Z, int(cz) + wz + 1)
        gx = ((np.arange(x0, x1) - cx) / rx) ** 2
        gy = ((np.arange(y0, y1) - cy) / ry) ** 2
        gz = ((np.arange(z0, z1) - cz) / rz) ** 2
        blob = 0.95 * np.exp(-0.5 * (gx[:, None, None] + gy[None, :, None] + gz[None, None, :]))
        np.maximum(prob[x0:x1, y0:y1, z0:z1], blob.astype(np.float32), out=prob[x0:x1, y0:y1, z0:z1])
    if speckle > 0:
        prob[rng.uniform(size=prob.shape) < speckle] = 0.9
    return prob


def load_ffn_npz(path) -> dict:
    """FFN weights stored flat (w1, w2, w3, b3, bn{1,2}_{gamma,beta,mean,var}; any float dtype) -> the nested float32 dict of
    make_ffn_weights.  The package's data/ffn_synthetic_trained.npz (made by tests/golden/train_synthetic_ffn.py) is such a file."""
    z = np.load(path)
    f = lambda k: np.asarray(z[k], dtype=np.float32)
    bn = lambda p: {k: f(f"{p}_{k}") for k in ("gamma", "beta", "mean", "var")}
    return {"w1": f("w1"), "bn1": bn("bn1"), "w2": f("w2"), "bn2": bn("bn2"), "w3": f("w3").reshape(FFN_HID, 1), "b3": f("b3").reshape(1)}


TRAINED_FFN_PATH = Path(__file__).resolve().parent / "data" / "ffn_synthetic_trained.npz"


def load_trained_ffn() -> dict:
    """The FFN trained on the reference's synthetic-pair recipe (ffn.py:18-53) that ships with the package (package data, not a test
    fixture: FrameChain.synthetic and bench.py use it as the default discriminating matcher)."""
    return load_ffn_npz(TRAINED_FFN_PATH)


def make_passthrough_unet_weights(arch_name: str = "unet3_a", seed: int = 0, gain: float = 4.0, bias: float = -3.0,
                                  noise: float = 0.02) -> dict:
    """U-Net weights that turn a normalised stack into a cell-like probability map, for tests of the *chained* frame
    (LCN -> U-Net -> regions -> match -> correction), where a Glorot-init net would only produce noise regions.

    Every conv block carries its first input channel through (centre tap 1 -> channel 0; identity BatchNorm statistics; the
    decoder reads channel 0 of the skip tensor), all other taps are small seeded noise so that no kernel path is trivially
    sparse; the head is sigmoid(gain * channel0 + bias).  Not a trained model: prob ~ sigmoid(gain * relu(lcn) + bias)."""
    arch = ARCHS[arch_name]
    rng = np.random.default_rng(seed + 77)
    layers = arch.conv_layers()
    # input channel that carries the signal into conv i: 0 everywhere, except decoder convs over concat([up(low), skip]),
    # where the skip tensor starts after the upsampled channels
    n_down = len(arch.down)
    signal_in = [0] * len(layers)
    c = arch.down[-1][1]
    for i in range(n_down):                       # first conv of decoder stage i (also the output stage)
        li = 2 * n_down + 2 * i
        signal_in[li] = 0 if i == 0 else arch.up[i - 1][1]
    signal_in[2 * n_down + 2 * n_down] = arch.up[-1][1] if len(layers) > 4 * n_down else 0
    convs = []
    for li, (cin, cout) in enumerate(layers):
        k = (rng.normal(0.0, noise / np.sqrt(27.0 * cin), (3, 3, 3, cin, cout))).astype(np.float32)
        # the bottleneck path (decoder stage 0 reads the pooled tensor) carries no fine signal: feed the skip instead
        k[1, 1, 1, signal_in[li], 0] = 1.0
        convs.append({"kernel": k, "bias": np.zeros(cout, np.float32), "gamma": np.ones(cout, np.float32),
                      "beta": np.zeros(cout, np.float32), "mean": np.zeros(cout, np.float32),
                      "var": np.full(cout, 1.0 - 1e-3, np.float32)})
    hk = (rng.normal(0.0, noise, (1, 1, 1, arch.out[1], 1))).astype(np.float32)
    hk[0, 0, 0, 0, 0] = gain
    return {"arch": arch_name, "convs": convs, "head": {"kernel": hk, "bias": np.array([bias], np.float32)}}


def make_legacy_frame_case(seed: int = 0, siz_xyz=(120, 136, 14), z_scaling: int = 5, z_xy_ratio: float = 4.0, n_cells: int = 40,
                           move: float = 2.5, margin: float = 10.0, edge_cells: int = 0):
    """Synthetic state for one frame of the legacy Tracker (tracker.py:1138-1175): a volume-1 label image on the
    z-interpolated grid (non-touching ellipsoids = what interpolate_seg leaves in seg_cells_interpolated_corrected), the raw
    uint16 stack of a later volume in which every cell has moved by a smooth field plus jitter, and a cell/background
    probability map of that stack (float16, the dtype of the reference's unet_cache files)."""
    rng = np.random.default_rng(seed)
    X, Y, Z = (int(v) for v in siz_xyz)
    ZI = Z * z_scaling
    rad = np.array([4.5, 4.5, 1.3])                                   # layer units
    centres = [np.array([X * (k + 1.0) / (edge_cells + 1.0), margin, Z / 2.0]) for k in range(edge_cells)]   # cells that will drift
    tries = 0                                                                                                # into the boundary zone
    while len(centres) < n_cells and tries < 20000:
        tries += 1
        c = rng.uniform([margin, margin, 2.0], [X - margin, Y - margin, Z - 2.0])
        if all(np.sum(((c - o) / (2.4 * rad)) ** 2) > 1.0 for o in centres):
            centres.append(c)
    centres = np.asarray(centres)
    n = len(centres)
    seg = np.zeros((X, Y, ZI), dtype=np.int32)
    for i, c in enumerate(centres):
        r = rad * rng.uniform(0.8, 1.1, 3)
        ci = np.array([c[0], c[1], c[2] * z_scaling + z_scaling // 2]); ri = np.array([r[0], r[1], r[2] * z_scaling])
        lo = np.maximum(np.floor(ci - ri).astype(int), 0); hi = np.minimum(np.ceil(ci + ri).astype(int) + 1, (X, Y, ZI))
        g = np.meshgrid(*(np.arange(lo[a], hi[a]) for a in range(3)), indexing="ij")
        inside = sum(((g[a] - ci[a]) / ri[a]) ** 2 for a in range(3)) <= 1.0
        sub = seg[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]]
        sub[inside & (sub == 0)] = i + 1
    # smooth displacement field (layer units) + jitter
    A = (rng.uniform(0, 1, (3, 3)) - 0.5) * 0.04
    ctr = np.array([X / 2, Y / 2, Z / 2])
    disp = (centres - ctr) @ A + rng.normal(0, 0.4, centres.shape) + np.array([move, -0.6 * move, 0.15])
    disp[:, 2] *= 0.3
    moved = centres + disp
    gx, gy, gz = np.meshgrid(np.arange(X), np.arange(Y), np.arange(Z), indexing="ij")
    raw = rng.normal(100.0, 20.0, (X, Y, Z))
    blob = np.zeros((X, Y, Z))
    amps = rng.uniform(500, 2000, n)
    for c, a in zip(moved, amps):
        e = np.exp(-0.5 * (((gx - c[0]) / 2.6) ** 2 + ((gy - c[1]) / 2.6) ** 2 + ((gz - c[2]) / 0.8) ** 2))
        raw += a * e
        blob = np.maximum(blob, e)
    raw = np.clip(raw, 0, 65535).astype(np.uint16)
    prob = (1.0 / (1.0 + np.exp(-(9.0 * blob - 4.0)))).astype(np.float16)      # > 0.5 inside ~1 sigma... a soft ellipsoid per cell
    return {"siz_xyz": (X, Y, Z), "z_scaling": int(z_scaling), "z_xy_ratio": float(z_xy_ratio), "seg_interp": seg,
            "centres_vol1": centres, "centres_moved": moved, "raw": raw, "prob_f16": prob}
