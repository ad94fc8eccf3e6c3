#!/usr/bin/env python3
"""Record per-block Keras outputs of the reference's own models -- the one pin this build cannot produce itself.

NOT runnable in the build image (tensorflow==2.11 is neither installed nor installable there); UNTESTED for that reason.  Run it
once on any machine that has the reference's environment (requirements.txt: tensorflow==2.11) and a checkout of the reference:

    python tests/golden/make_keras_layer_outputs.py /path/to/3DeeCellTracker

and commit the resulting tests/golden/keras_layer_outputs.npz (~7 MB).  tests/test_keras_pin.py then holds oracle/unet_ref.py,
oracle/match_ref.py (CPU) and the HIP kernels (-m gpu) to what Keras itself computed, which upgrades every "oracle-unpinned"
row of DESIGN.md section 2 (U-Net conv / activation / BatchNorm / pool / upsample / concat order, FFN dense + BatchNorm, lcn_gpu).

What it does: builds unet3_a / unet3_c through the reference's `CellTracker.unet3d.unet3_a()` / `unet3_c()` and the FFN through
`CellTracker.ffn.FFN()`, assigns the seeded synthetic weights of 3deecelltracker_amd/synth.py (pure numpy, same seeds as the
tests) layer by layer, and records for a seeded input: the output of every BatchNormalization layer (= every conv block), the
probability map, the FFN scores of 4096 seeded feature rows, and lcn_gpu of a seeded image.  Inputs are regenerated from the
seeds by the test; only Keras outputs are stored (float16 for the big block outputs would lose the comparison: float32 kept,
blocks stored on a 2x2x1-strided sub-grid to bound the size)."""
import importlib
import sys
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(REPO))
sys.path.insert(0, sys.argv[1] if len(sys.argv) > 1 else ".")
synth = importlib.import_module("3deecelltracker_amd.synth")
ARCHS = importlib.import_module("3deecelltracker_amd.arch").ARCHS


def main():
    import tensorflow as tf
    from tensorflow.keras.layers import BatchNormalization, Conv3D
    from tensorflow.keras.models import Model
    ref_unet3d = importlib.import_module("CellTracker.unet3d")
    ref_ffn = importlib.import_module("CellTracker.ffn")
    ref_pre = importlib.import_module("CellTracker.preprocess")
    out = {"tf_version": np.array(tf.__version__)}
    for name in ("unet3_a", "unet3_c"):
        arch = ARCHS[name]
        w = synth.make_unet_weights(name, seed=1)
        model = getattr(ref_unet3d, name)()
        convs = [l for l in model.layers if isinstance(l, Conv3D)]
        bns = [l for l in model.layers if isinstance(l, BatchNormalization)]
        assert len(convs) == len(w["convs"]) + 1 and len(bns) == len(w["convs"])
        for l, src in zip(convs[:-1], w["convs"]):
            l.set_weights([src["kernel"], src["bias"]])
        for l, src in zip(bns, w["convs"]):
            l.set_weights([src["gamma"], src["beta"], src["mean"], src["var"]])
        convs[-1].set_weights([w["head"]["kernel"], w["head"]["bias"]])
        patch = np.random.default_rng(2).normal(size=arch.input_shape).astype(np.float32)
        probe = Model(inputs=model.input, outputs=[l.output for l in bns] + [model.output])
        res = probe.predict(patch[None, :, :, :, None])
        for i, r in enumerate(res[:-1]):
            out[f"{name}_block{i}"] = np.asarray(r[0, ::2, ::2, :, :], dtype=np.float32)
        out[f"{name}_prob"] = np.asarray(res[-1][0, :, :, :, 0], dtype=np.float32)
    fw = synth.make_ffn_weights(seed=0, gain=6.0, shift=-3.0)
    ffn = ref_ffn.FFN()
    x = np.random.default_rng(3).normal(size=(4096, 122)).astype(np.float32)
    ffn(x[:2])                                                          # build the variables
    d1, b1 = ffn.feat_layer1.layers[0], ffn.feat_layer1.layers[1]
    d2, b2 = ffn.combine_feat2.layers[0], ffn.combine_feat2.layers[1]
    d1.set_weights([fw["w1"]]); b1.set_weights([fw["bn1"][k] for k in ("gamma", "beta", "mean", "var")])
    d2.set_weights([fw["w2"]]); b2.set_weights([fw["bn2"][k] for k in ("gamma", "beta", "mean", "var")])
    ffn.pred.layers[0].set_weights([fw["w3"], fw["b3"]])
    out["ffn_scores"] = np.asarray(ffn.predict(x, batch_size=1024), dtype=np.float32)[:, 0]
    img = np.random.default_rng(21).integers(0, 3000, size=(64, 64, 16)).astype(np.float64)
    out["lcn_gpu"] = np.asarray(ref_pre.lcn_gpu(img, 100.0, filter_size=(27, 27, 1)), dtype=np.float32)
    np.savez_compressed(Path(__file__).resolve().parent / "keras_layer_outputs.npz", **out)
    print("written", Path(__file__).resolve().parent / "keras_layer_outputs.npz")


if __name__ == "__main__":
    main()
