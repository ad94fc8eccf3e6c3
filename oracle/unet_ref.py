"""ORACLE (test infrastructure -- never imported by the product path).

CPU restatement in numpy of the reference's 3D U-Net sliding-window inference:

* network wiring           <- reference CellTracker/unet3d.py:84-98 (_unet3_depth3), :40-67 (unet3_b),
                              blocks :101-200 (conv->activation->BN, [conv,conv]->pool,
                              [conv,conv]->upsample->concat[up, skip])
* sliding-window predictor <- reference CellTracker/unet3d.py:203-279 (unet3_prediction,
                              _get_sizes_padded_im)

PARITY STATUS.  The tiler half (pad / patch grid / centre-crop stitch) is PINNED: it is checked
against golden vectors produced by running the reference's own ``unet3_prediction`` with fake
models (tests/golden/make_golden.py).  The layer arithmetic is "parity unpinned": every
Conv3D / BatchNormalization / MaxPooling3D / UpSampling3D forward pass of the reference executes
inside tensorflow==2.11 (requirements.txt:4), which is neither under /root/reference nor installed
here, and the reference has no tests or recorded outputs for it.  The arithmetic below restates
the published Keras layer semantics (see 3deecelltracker_amd/arch.py) and is cross-checked against
an independent torch-CPU float64 evaluation in tests/test_oracle_unet.py (torch is not the
reference; it only guards against restatement bugs).
"""
from __future__ import annotations

import math

import numpy as np

LEAKY_ALPHA = 0.3
BN_EPS = 1e-3


# --------------------------------------------------------------------------- layers
def conv3d_same(x: np.ndarray, kernel: np.ndarray, bias: np.ndarray) -> np.ndarray:
    """x: (X,Y,Z,Cin); kernel: (3,3,3,Cin,Cout) cross-correlation, zero 'same' padding."""
    X, Y, Z, cin = x.shape
    cout = kernel.shape[-1]
    xp = np.zeros((X + 2, Y + 2, Z + 2, cin), dtype=x.dtype)
    xp[1:-1, 1:-1, 1:-1, :] = x
    acc = np.zeros((X * Y * Z, cout), dtype=x.dtype)
    for a in range(3):
        for b in range(3):
            for c in range(3):
                win = xp[a:a + X, b:b + Y, c:c + Z, :].reshape(-1, cin)
                acc += win @ kernel[a, b, c].astype(x.dtype)
    acc += bias.astype(x.dtype)[None, :]
    return acc.reshape(X, Y, Z, cout)


def activation(x: np.ndarray, act: int) -> np.ndarray:
    if act == 0:  # LeakyReLU(0.3)
        return np.where(x >= 0, x, x * x.dtype.type(LEAKY_ALPHA))
    return np.maximum(x, 0)


def batchnorm(x: np.ndarray, layer: dict) -> np.ndarray:
    dt = x.dtype
    inv = layer["gamma"].astype(dt) / np.sqrt(layer["var"].astype(dt) + dt.type(BN_EPS))
    return (x - layer["mean"].astype(dt)) * inv + layer["beta"].astype(dt)


def conv_block(x, layer, act):
    """conv -> activation -> BN (BN is AFTER the non-linearity: unet3d.py:117-119, :139-140)."""
    return batchnorm(activation(conv3d_same(x, layer["kernel"], layer["bias"]), act), layer)


def maxpool(x: np.ndarray, pool) -> np.ndarray:
    X, Y, Z, C = x.shape
    px, py, pz = pool
    x = x[:X // px * px, :Y // py * py, :Z // pz * pz]
    return x.reshape(X // px, px, Y // py, py, Z // pz, pz, C).max(axis=(1, 3, 5))


def upsample(x: np.ndarray, size) -> np.ndarray:
    for ax, r in enumerate(size):
        if r > 1:
            x = np.repeat(x, r, axis=ax)
    return x


def unet_forward(patch: np.ndarray, weights: dict, arch, dtype=np.float32, collect=None) -> np.ndarray:
    """patch: (X,Y,Z) or (X,Y,Z,1) -> probability map (X,Y,Z).  `collect` (list) receives every
    intermediate conv-block output, in execution order, for layer-wise parity tests."""
    x = np.asarray(patch, dtype=dtype)
    if x.ndim == 3:
        x = x[..., None]
    convs = weights["convs"]
    i = 0
    skips = []
    for _ in arch.down:
        x = conv_block(x, convs[i], arch.act); i += 1
        if collect is not None: collect.append(x)
        x = conv_block(x, convs[i], arch.act); i += 1
        if collect is not None: collect.append(x)
        skips.append(x)
        x = maxpool(x, arch.pool)
    for _ in arch.up:
        x = conv_block(x, convs[i], arch.act); i += 1
        if collect is not None: collect.append(x)
        x = conv_block(x, convs[i], arch.act); i += 1
        if collect is not None: collect.append(x)
        x = np.concatenate([upsample(x, arch.pool), skips.pop()], axis=-1)
    for _ in range(2):
        x = conv_block(x, convs[i], arch.act); i += 1
        if collect is not None: collect.append(x)
    head = weights["head"]
    logit = x.reshape(-1, x.shape[-1]) @ head["kernel"].reshape(-1, 1).astype(dtype) + head["bias"].astype(dtype)
    prob = 1.0 / (1.0 + np.exp(-logit))
    return prob.reshape(x.shape[:3]).astype(dtype)


def unet_forward_torch(patch: np.ndarray, weights: dict, arch, dtype=np.float32, threads: int | None = None) -> np.ndarray:
    """The same network through torch's CPU conv3d (oneDNN, all host threads): an independent evaluation used (fp64) to
    guard the numpy restatement above against restatement bugs, and (fp32) as the multi-threaded CPU baseline that
    SURVEY 8d asks for -- the numpy shifted-matmul conv above is bound by skinny GEMMs and far slower.  torch is not the
    reference; the Keras layer semantics are the ones listed in 3deecelltracker_amd/arch.py."""
    import torch
    import torch.nn.functional as F
    if threads:
        torch.set_num_threads(int(threads))
    td = torch.float64 if np.dtype(dtype) == np.float64 else torch.float32
    t = lambda a: torch.tensor(np.asarray(a), dtype=td)
    x = t(np.asarray(patch).reshape(np.asarray(patch).shape[:3]))[None, None]          # N C X Y Z

    def block(x, layer):
        k = t(layer["kernel"]).permute(4, 3, 0, 1, 2)                                  # Cout Cin kx ky kz
        y = F.conv3d(x, k, t(layer["bias"]), padding=1)
        y = F.leaky_relu(y, LEAKY_ALPHA) if arch.act == 0 else F.relu(y)
        sh = (1, -1, 1, 1, 1)
        return (y - t(layer["mean"]).view(sh)) / torch.sqrt(t(layer["var"]).view(sh) + BN_EPS) * \
            t(layer["gamma"]).view(sh) + t(layer["beta"]).view(sh)
    convs = weights["convs"]; i = 0; skips = []
    with torch.no_grad():
        for _ in arch.down:
            x = block(x, convs[i]); i += 1
            x = block(x, convs[i]); i += 1
            skips.append(x)
            x = F.max_pool3d(x, arch.pool)
        for _ in arch.up:
            x = block(x, convs[i]); i += 1
            x = block(x, convs[i]); i += 1
            x = torch.cat([F.interpolate(x, scale_factor=tuple(float(p) for p in arch.pool), mode="nearest"), skips.pop()], 1)
        for _ in range(2):
            x = block(x, convs[i]); i += 1
        k = t(weights["head"]["kernel"]).permute(4, 3, 0, 1, 2)
        return torch.sigmoid(F.conv3d(x, k, t(weights["head"]["bias"])))[0, 0].numpy()


def unet3_prediction_torch(vol_xyz: np.ndarray, weights: dict, arch, shrink=(24, 24, 2), dtype=np.float64, batch: int = 4):
    """Whole sliding-window prediction (unet3d.py:203-255) with EVERY patch evaluated by unet_forward_torch's network, `batch` patches
    per conv call -> stitched (x, y, z) float32 map.  What the full-size GPU parity tests compare all 75 / 18 / 88 patches against."""
    import torch
    import torch.nn.functional as F
    td = torch.float64 if np.dtype(dtype) == np.float64 else torch.float32
    t = lambda a: torch.tensor(np.asarray(a), dtype=td)
    plan = tile_plan(vol_xyz.shape, arch.input_shape, arch.input_shape, shrink)
    patches = gather_patches(np.asarray(vol_xyz), plan)
    convs = weights["convs"]
    packed = [(t(l["kernel"]).permute(4, 3, 0, 1, 2).contiguous(), t(l["bias"]),
               (t(l["gamma"]) / torch.sqrt(t(l["var"]) + BN_EPS)).view(1, -1, 1, 1, 1), t(l["mean"]).view(1, -1, 1, 1, 1),
               t(l["beta"]).view(1, -1, 1, 1, 1)) for l in convs]
    head_k = t(weights["head"]["kernel"]).permute(4, 3, 0, 1, 2); head_b = t(weights["head"]["bias"])

    def block(x, i):
        k, b, inv, mean, beta = packed[i]
        y = F.conv3d(x, k, b, padding=1)
        y = F.leaky_relu(y, LEAKY_ALPHA) if arch.act == 0 else F.relu(y)
        return (y - mean) * inv + beta
    pred = np.empty(patches.shape, dtype=np.float32)
    with torch.no_grad():
        for s in range(0, len(patches), batch):
            x = t(patches[s:s + batch])[:, None]
            i = 0; skips = []
            for _ in arch.down:
                x = block(x, i); x = block(x, i + 1); i += 2
                skips.append(x)
                x = F.max_pool3d(x, arch.pool)
            for _ in arch.up:
                x = block(x, i); x = block(x, i + 1); i += 2
                x = torch.cat([F.interpolate(x, scale_factor=tuple(float(p) for p in arch.pool), mode="nearest"), skips.pop()], 1)
            x = block(x, i); x = block(x, i + 1)
            pred[s:s + batch] = torch.sigmoid(F.conv3d(x, head_k, head_b))[:, 0].numpy()
    return scatter_centres(pred, plan, vol_xyz.shape), pred, plan


# --------------------------------------------------------------------------- tiler
def padded_size(img_size: int, centre: int):
    """unet3d.py:259-279: number of sub-regions and the padded extent along one axis."""
    num = int(math.ceil(img_size * 1.0 / centre))
    return num * centre, num


def reflect_index(i: np.ndarray, n: int) -> np.ndarray:
    """Source index for numpy 'reflect' padding (edge not repeated), valid for any pad width."""
    if n == 1:
        return np.zeros_like(i)
    p = 2 * (n - 1)
    t = np.mod(i, p)
    return np.where(t >= n, p - t, t)


def tile_plan(vol_shape, net_in, net_out, shrink):
    """Patch origins (in padded coordinates), centre size, grid, padded extents."""
    centre = tuple(net_out[a] - 2 * shrink[a] for a in range(3))
    ext, grid = zip(*(padded_size(vol_shape[a], centre[a]) for a in range(3)))
    before = tuple(shrink)
    after = tuple(shrink[a] + ext[a] - vol_shape[a] for a in range(3))
    return {"centre": centre, "grid": tuple(grid), "ext": tuple(ext), "before": before, "after": after,
            "net_in": tuple(net_in)}


def gather_patches(vol: np.ndarray, plan) -> np.ndarray:
    """vol (x,y,z) -> (P, nx, ny, nz) patches of the reflect-padded volume, P ordered like
    itertools.product(range(gx), range(gy), range(gz)) (unet3d.py:246)."""
    gx, gy, gz = plan["grid"]; cx, cy, cz = plan["centre"]; nx, ny, nz = plan["net_in"]
    bx, by, bz = plan["before"]
    out = np.empty((gx * gy * gz, nx, ny, nz), dtype=vol.dtype)
    p = 0
    for i in range(gx):
        ix = reflect_index(np.arange(i * cx, i * cx + nx) - bx, vol.shape[0])
        for j in range(gy):
            iy = reflect_index(np.arange(j * cy, j * cy + ny) - by, vol.shape[1])
            for k in range(gz):
                iz = reflect_index(np.arange(k * cz, k * cz + nz) - bz, vol.shape[2])
                out[p] = vol[np.ix_(ix, iy, iz)]
                p += 1
    return out


def scatter_centres(pred: np.ndarray, plan, vol_shape) -> np.ndarray:
    """pred (P, nx, ny, nz) -> stitched (x,y,z): centre crop of every patch (unet3d.py:237-255)."""
    gx, gy, gz = plan["grid"]; cx, cy, cz = plan["centre"]; bx, by, bz = plan["before"]
    ext = plan["ext"]
    full = np.zeros(ext, dtype=np.float32)
    p = 0
    for i in range(gx):
        for j in range(gy):
            for k in range(gz):
                full[i * cx:(i + 1) * cx, j * cy:(j + 1) * cy, k * cz:(k + 1) * cz] = \
                    pred[p, bx:bx + cx, by:by + cy, bz:bz + cz]
                p += 1
    return full[:vol_shape[0], :vol_shape[1], :vol_shape[2]]


def unet3_prediction_ref(img: np.ndarray, predict_patch, net_in, net_out=None, shrink=(24, 24, 2)) -> np.ndarray:
    """img (1,x,y,z,1) -> float32 (1,x,y,z,1).  `predict_patch(patch_xyz) -> prob_xyz`."""
    net_out = net_in if net_out is None else net_out
    vol = np.asarray(img)[0, :, :, :, 0]
    plan = tile_plan(vol.shape, net_in, net_out, shrink)
    patches = gather_patches(vol, plan)
    pred = np.stack([np.asarray(predict_patch(p), dtype=np.float32) for p in patches])
    return scatter_centres(pred, plan, vol.shape)[None, :, :, :, None]
