"""Quick timing of the U-Net volume path (dev helper, not the contract bench)."""
import importlib, sys, time
import numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
synth = importlib.import_module("3deecelltracker_amd.synth")
unet3d = importlib.import_module("3deecelltracker_amd.unet3d")
arch = importlib.import_module("3deecelltracker_amd.arch").UNET3_A
shape = tuple(int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (512, 512, 32)))
model = unet3d.unet3_a().set_weights_dict(synth.make_unet_weights("unet3_a", 0))
vol = torch.randn(*shape, device="cuda")
out = torch.zeros_like(vol)
for _ in range(2):
    model.predict_volume_device(vol, out=out)
torch.cuda.synchronize()
t0 = time.perf_counter(); K = 5
for _ in range(K):
    model.predict_volume_device(vol, out=out)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
centre, grid = unet3d.tile_plan(shape, arch.input_shape, (24, 24, 2))
npatch = grid[0] * grid[1] * grid[2]
print(f"shape {shape} patches {npatch}: {dt*1e3:.2f} ms/vol  {1/dt:.2f} vol/s  {npatch*arch.flops_per_patch()/dt/1e12:.1f} TFLOP/s fp32")
