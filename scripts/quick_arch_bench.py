"""Patch throughput of the three U-Net architectures (dev helper)."""
import importlib, sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
synth = importlib.import_module("3deecelltracker_amd.synth"); unet3d = importlib.import_module("3deecelltracker_amd.unet3d")
archs = importlib.import_module("3deecelltracker_amd.arch").ARCHS
for name, nb in (("unet3_a", 75), ("unet3_c", 150), ("unet3_b", 24)):
    arch = archs[name]
    model = getattr(unet3d, name)().set_weights_dict(synth.make_unet_weights(name, 0))
    x = torch.randn(nb, *arch.input_shape, device="cuda")
    for _ in range(2): model.predict_device(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): model.predict_device(x)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f"{name}: {nb} patches {dt*1e3:.2f} ms  {nb*arch.flops_per_patch()/dt/1e12:.1f} TFLOP/s  ({dt/nb*1e3:.3f} ms/patch)")
