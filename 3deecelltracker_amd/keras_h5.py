"""Keras `.h5` weight import for the U-Nets (`model.save_weights` / `model.save` files, reference
tracker.py:579, unet3d.py TrainingUNet3D) and the FFN (reference trackerlite.py:57-63, ffn.py TrainFFN).

Needs h5py, which is NOT part of this image: the module is imported lazily and raises ImportError otherwise, so it is
untested here ("parity unpinned"); `.npz` files written by `save_weights` of the mirrors are the native format.

Keras HDF5 layout (Keras 2.x): weights live under the root (weights-only file) or under `model_weights` (full model);
every layer group carries a `weight_names` attribute listing its datasets in creation order, the parent carries
`layer_names`.
"""
from __future__ import annotations

import numpy as np


def _iter_weights(path):
    import h5py
    out = []
    with h5py.File(path, "r") as f:
        root = f["model_weights"] if "model_weights" in f else f
        names = [n.decode() if isinstance(n, bytes) else n for n in root.attrs.get("layer_names", list(root.keys()))]
        for ln in names:
            grp = root[ln]
            wn = [n.decode() if isinstance(n, bytes) else n for n in grp.attrs.get("weight_names", [])]
            for w in wn:
                out.append((w, np.asarray(grp[w])))
    return out


def read_unet_h5(path, arch) -> dict:
    """-> the dict container of synth.make_unet_weights (Keras layouts are kept as they are)."""
    ws = _iter_weights(path)
    kernels = [(n, a) for n, a in ws if a.ndim == 5]
    biases = {n.rsplit("/", 1)[0]: a for n, a in ws if n.endswith("bias:0")}
    bn = {}
    for n, a in ws:
        layer, leaf = n.rsplit("/", 1)
        if leaf.split(":")[0] in ("gamma", "beta", "moving_mean", "moving_variance"):
            bn.setdefault(layer, {})[leaf.split(":")[0]] = a
    bn_layers = list(bn.values())
    layers = arch.conv_layers()
    if len(kernels) != len(layers) + 1 or len(bn_layers) != len(layers):
        raise ValueError(f"{path}: found {len(kernels)} conv kernels / {len(bn_layers)} BatchNorm layers, expected "
                         f"{len(layers) + 1} / {len(layers)} for {arch.name}")
    convs = []
    for (kname, k), b, (cin, cout) in zip(kernels[:-1], bn_layers, layers):
        if k.shape != (3, 3, 3, cin, cout):
            raise ValueError(f"{path}: kernel {kname} has shape {k.shape}, expected {(3, 3, 3, cin, cout)}")
        convs.append({"kernel": k.astype(np.float32), "bias": biases[kname.rsplit("/", 1)[0]].astype(np.float32),
                      "gamma": b["gamma"].astype(np.float32), "beta": b["beta"].astype(np.float32),
                      "mean": b["moving_mean"].astype(np.float32), "var": b["moving_variance"].astype(np.float32)})
    hname, hk = kernels[-1]
    if hk.shape != (1, 1, 1, arch.out[1], 1):
        raise ValueError(f"{path}: head kernel has shape {hk.shape}")
    return {"arch": arch.name, "convs": convs,
            "head": {"kernel": hk.astype(np.float32), "bias": biases[hname.rsplit("/", 1)[0]].astype(np.float32)}}


def read_ffn_h5(path) -> dict:
    """-> the dict container of synth.make_ffn_weights."""
    ws = _iter_weights(path)
    dense = [a for n, a in ws if n.endswith("kernel:0")]
    by_shape = {a.shape: a for a in dense}
    bn, order = {}, []
    for n, a in ws:
        layer, leaf = n.rsplit("/", 1)
        key = leaf.split(":")[0]
        if key in ("gamma", "beta", "moving_mean", "moving_variance"):
            if layer not in bn:
                order.append(layer)
            bn.setdefault(layer, {})[key] = a.astype(np.float32)
    b3 = [a for n, a in ws if n.endswith("bias:0") and a.shape == (1,)]
    if (61, 512) not in by_shape or (1024, 512) not in by_shape or (512, 1) not in by_shape or len(order) != 2 or not b3:
        raise ValueError(f"{path}: not an FFN weight file (dense kernels {sorted(by_shape)}, {len(order)} BatchNorm layers)")

    def pack(b):
        return {"gamma": b["gamma"], "beta": b["beta"], "mean": b["moving_mean"], "var": b["moving_variance"]}
    return {"w1": by_shape[(61, 512)].astype(np.float32), "bn1": pack(bn[order[0]]),
            "w2": by_shape[(1024, 512)].astype(np.float32), "bn2": pack(bn[order[1]]),
            "w3": by_shape[(512, 1)].astype(np.float32), "b3": b3[0].astype(np.float32)}
