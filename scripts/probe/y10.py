"""4 x 10 x 16 conv tiles (CT_CONV_Y10 = bit mask of conv indices) against the shipped 4 x 8 x 16 ones, layer by layer.

Every mask runs in a child process (the switch is read once): the child dumps the conv blocks and the probability map of one seeded patch
(parity against mask 0: the tile shape must not change a value beyond the summation order -- it does not change that either, every output
still sums its 27 x Cin products in the same order) and prints the per-layer times of the 512 x 512 x 32 volume.

    python scripts/probe/y10.py [mask ...]          (default: 0, every single bit 2..13, all bits)
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CHILD = r"""
import ctypes, importlib, sys
import numpy as np, torch
sys.path.insert(0, sys.argv[1])
out = sys.argv[2]
synth = importlib.import_module("3deecelltracker_amd.synth")
unet3d = importlib.import_module("3deecelltracker_amd.unet3d")
_lib = importlib.import_module("3deecelltracker_amd._lib")
arch = importlib.import_module("3deecelltracker_amd.arch").ARCHS["unet3_a"]
model = unet3d.unet3_a().set_weights_dict(synth.make_unet_weights("unet3_a", seed=3))
patches = np.random.default_rng(4).normal(size=(3,) + tuple(arch.input_shape)).astype(np.float32)
got, dump = model.predict_device(torch.from_numpy(patches[:1]).cuda(), layer_dump=True)
both = model.predict_device(torch.from_numpy(patches).cuda())
vol = torch.from_numpy(np.random.default_rng(5).normal(size=(200, 330, 21)).astype(np.float32)).cuda()
pv = torch.zeros_like(vol); model.predict_volume_device(vol, out=pv)
torch.cuda.synchronize()
np.savez(out, prob=both.cpu().numpy(), dump=dump.cpu().numpy(), vol=pv.cpu().numpy())
vol = torch.randn(512, 512, 32, device="cuda"); o = torch.zeros_like(vol)
for _ in range(3): model.predict_volume_device(vol, out=o)
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): model.predict_volume_device(vol, out=o)
e1.record(); torch.cuda.synchronize()
print("VOL", e0.elapsed_time(e1) / 10)
L = _lib.lib(); h = model._handle
L.ct_unet_set_timing(h, 1)
for _ in range(5): model.predict_volume_device(vol, out=o)
nl = L.ct_unet_num_conv_layers(h)
ms = (ctypes.c_float * nl)(); cnt = (ctypes.c_int * nl)()
L.ct_unet_get_timing(h, ms, cnt, nl)
print("LAYERS", " ".join(f"{ms[i] / 5:.4f}" for i in range(nl)))
"""


def main():
    masks = [int(v, 0) for v in sys.argv[1:]] or [0] + [1 << b for b in range(2, 14)] + [0x3ffc]
    res, ref = {}, None
    with tempfile.TemporaryDirectory() as td:
        for m in masks:
            out = os.path.join(td, f"m{m}.npz")
            env = dict(os.environ, CT_CONV_Y10=hex(m))
            r = subprocess.run([sys.executable, "-c", CHILD, ROOT, out], env=env, capture_output=True, text=True, timeout=900)
            if r.returncode:
                print(f"mask {m:#x}: FAILED\n{r.stderr[-1500:]}"); continue
            z = np.load(out)
            if ref is None:
                ref = {k: z[k] for k in z.files}
            vol = [ln for ln in r.stdout.splitlines() if ln.startswith("VOL")][0].split()[1]
            lay = [float(v) for v in [ln for ln in r.stdout.splitlines() if ln.startswith("LAYERS")][0].split()[1:]]
            errs = {k: float(np.abs(z[k] - ref[k]).max()) for k in ref}
            res[m] = (float(vol), lay, errs)
            print(f"mask {m:#06x}: {float(vol):.3f} ms/vol   max|diff| to mask 0: dump {errs['dump']:.3g} prob {errs['prob']:.3g} volume {errs['vol']:.3g}", flush=True)
    if 0 in res:
        base = res[0][1]
        print("per-layer ms (mask 0):", " ".join(f"L{i}={v:.3f}" for i, v in enumerate(base)))
        for m, (vol, lay, _) in res.items():
            if m == 0:
                continue
            ch = [f"L{i}: {base[i]:.3f} -> {lay[i]:.3f}" for i in range(len(lay)) if (m >> i) & 1]
            print(f"mask {m:#06x}: " + "; ".join(ch))


if __name__ == "__main__":
    main()
