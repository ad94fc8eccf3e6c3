"""Keras `.h5` weight import for the U-Nets (`model.save_weights` / `model.save` files, reference
tracker.py:579, unet3d.py TrainingUNet3D) and the FFN (reference trackerlite.py:57-63, ffn.py TrainFFN).

Needs h5py, which the image's main interpreter lacks: the module is imported lazily.  tests/test_keras_h5.py writes files in the Keras 2.x
layout (weights-only and full-model, flat for the functional U-Nets, nested Sequential groups for the subclassed FFN) and reads
them back wherever h5py is importable -- on this image under /opt/conda/bin/python3.9 (h5py 3.3), in a subprocess; no real Keras-written
file has been available ("parity unpinned").  `.npz`
files written by `save_weights` of the mirrors are the native format.

Keras HDF5 layout (Keras 2.x `save_weights_to_hdf5_group`): weights live under the root (weights-only file) or under
`model_weights` (full model); the parent carries `layer_names` (model.layers order), every layer group a `weight_names`
attribute listing its datasets (paths relative to the layer group, e.g. `conv3d_3/kernel:0`, or `dense/kernel:0` inside the
group of a nested Sequential).
"""
from __future__ import annotations

import numpy as np

_BN_KEYS = ("gamma", "beta", "moving_mean", "moving_variance")


def _iter_weights(path):
    """-> [(layer group name, weight name, array)] in file order (layer_names x weight_names)."""
    import h5py
    out = []
    with h5py.File(path, "r") as f:
        root = f["model_weights"] if "model_weights" in f else f
        names = [n.decode() if isinstance(n, bytes) else n for n in root.attrs.get("layer_names", list(root.keys()))]
        for ln in names:
            grp = root[ln]
            wn = [n.decode() if isinstance(n, bytes) else n for n in grp.attrs.get("weight_names", [])]
            for w in wn:
                out.append((ln, w, np.asarray(grp[w])))
    return out


def _leaf(name):
    return name.rsplit("/", 1)[-1].split(":")[0]


def _owner(name):
    return name.rsplit("/", 1)[0] if "/" in name else ""


def _bn_pack(b, where):
    missing = [k for k in _BN_KEYS if k not in b]
    if missing:
        raise ValueError(f"{where}: BatchNormalization weights {missing} are missing")
    return {"gamma": b["gamma"].astype(np.float32), "beta": b["beta"].astype(np.float32),
            "mean": b["moving_mean"].astype(np.float32), "var": b["moving_variance"].astype(np.float32)}


def read_unet_h5(path, arch) -> dict:
    """-> the dict container of synth.make_unet_weights (Keras layouts are kept as they are).

    Conv and BatchNormalization layers are paired by position in the file's layer order AND checked by shape: every 3x3x3 conv
    must be followed, before the next conv, by exactly one BatchNormalization of its output width (unet3d.py:117-119,
    :139-140); the 1x1x1 head has none.  Anything else raises instead of loading weights into the wrong layer."""
    ws = _iter_weights(path)
    blocks = []                         # [{"kernel", "bias", "bn": {...}}] in layer order
    for ln, wn, a in ws:
        leaf = _leaf(wn)
        if leaf == "kernel":
            if a.ndim != 5:
                raise ValueError(f"{path}: unexpected kernel {ln}/{wn} of shape {a.shape} in a 3D U-Net file")
            blocks.append({"name": f"{ln}/{wn}", "kernel": a, "bias": None, "bn": {}, "owner": (ln, _owner(wn))})
        elif leaf == "bias":
            if not blocks or blocks[-1]["owner"] != (ln, _owner(wn)) or blocks[-1]["bias"] is not None:
                raise ValueError(f"{path}: bias {ln}/{wn} does not follow its kernel")
            blocks[-1]["bias"] = a
        elif leaf in _BN_KEYS:
            if not blocks:
                raise ValueError(f"{path}: BatchNormalization {ln}/{wn} precedes every conv layer")
            if leaf in blocks[-1]["bn"]:
                raise ValueError(f"{path}: two BatchNormalization layers after conv {blocks[-1]['name']}")
            blocks[-1]["bn"][leaf] = a
    layers = arch.conv_layers()
    if len(blocks) != len(layers) + 1:
        raise ValueError(f"{path}: found {len(blocks)} conv layers, expected {len(layers) + 1} for {arch.name}")
    convs = []
    for blk, (cin, cout) in zip(blocks[:-1], layers):
        if blk["kernel"].shape != (3, 3, 3, cin, cout):
            raise ValueError(f"{path}: kernel {blk['name']} has shape {blk['kernel'].shape}, expected {(3, 3, 3, cin, cout)}")
        if blk["bias"] is None or blk["bias"].shape != (cout,):
            raise ValueError(f"{path}: conv {blk['name']} has no bias of width {cout}")
        bn = _bn_pack(blk["bn"], f"{path}: conv {blk['name']}")
        if any(v.shape != (cout,) for v in bn.values()):
            raise ValueError(f"{path}: the BatchNormalization after {blk['name']} is not {cout} wide")
        convs.append(dict({"kernel": blk["kernel"].astype(np.float32), "bias": blk["bias"].astype(np.float32)}, **bn))
    head = blocks[-1]
    if head["kernel"].shape != (1, 1, 1, arch.out[1], 1) or head["bn"] or head["bias"] is None:
        raise ValueError(f"{path}: the last conv {head['name']} is not the 1x1x1 sigmoid head (shape {head['kernel'].shape})")
    return {"arch": arch.name, "convs": convs,
            "head": {"kernel": head["kernel"].astype(np.float32), "bias": head["bias"].astype(np.float32)}}


def read_ffn_h5(path) -> dict:
    """-> the dict container of synth.make_ffn_weights.  The FFN is a subclassed Model of three Sequential groups
    (ffn.py:237-258): each BatchNormalization is taken from the group that holds the Dense kernel it normalises."""
    ws = _iter_weights(path)
    groups = {}
    for ln, wn, a in ws:
        g = groups.setdefault(ln, {"kernels": [], "bias": [], "bn": {}})
        leaf = _leaf(wn)
        if leaf == "kernel":
            g["kernels"].append(a)
        elif leaf == "bias":
            g["bias"].append(a)
        elif leaf in _BN_KEYS:
            if leaf in g["bn"]:
                raise ValueError(f"{path}: group {ln} holds more than one BatchNormalization layer")
            g["bn"][leaf] = a

    def find(shape):
        hits = [(ln, g) for ln, g in groups.items() if any(k.shape == shape for k in g["kernels"])]
        if len(hits) != 1:
            raise ValueError(f"{path}: expected exactly one layer group with a Dense kernel of shape {shape}, found {len(hits)}")
        ln, g = hits[0]
        return ln, g, [k for k in g["kernels"] if k.shape == shape][0]
    n1, g1, w1 = find((61, 512))
    n2, g2, w2 = find((1024, 512))
    n3, g3, w3 = find((512, 1))
    if len({n1, n2, n3}) != 3:
        raise ValueError(f"{path}: the three Dense layers should live in three Sequential groups, found {[n1, n2, n3]}")
    b3 = [b for b in g3["bias"] if b.shape == (1,)]
    if len(b3) != 1 or g1["bias"] or g2["bias"]:
        raise ValueError(f"{path}: not an FFN weight file (bias layout)")
    bn1, bn2 = _bn_pack(g1["bn"], f"{path}: group {n1}"), _bn_pack(g2["bn"], f"{path}: group {n2}")
    if any(v.shape != (512,) for v in list(bn1.values()) + list(bn2.values())):
        raise ValueError(f"{path}: BatchNormalization layers are not 512 wide")
    return {"w1": w1.astype(np.float32), "bn1": bn1, "w2": w2.astype(np.float32), "bn2": bn2,
            "w3": w3.astype(np.float32), "b3": b3[0].astype(np.float32)}
