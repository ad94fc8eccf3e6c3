"""unet3_b alone (dev helper for profiling)."""
import importlib, sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
synth = importlib.import_module("3deecelltracker_amd.synth"); unet3d = importlib.import_module("3deecelltracker_amd.unet3d")
arch = importlib.import_module("3deecelltracker_amd.arch").ARCHS["unet3_b"]
model = unet3d.unet3_b().set_weights_dict(synth.make_unet_weights("unet3_b", 0))
x = torch.randn(24, *arch.input_shape, device="cuda")
for _ in range(2): model.predict_device(x)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): model.predict_device(x)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print(f"unet3_b: 24 patches {dt*1e3:.2f} ms  {24*arch.flops_per_patch()/dt/1e12:.1f} TFLOP/s")
