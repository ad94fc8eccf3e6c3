"""Host-buffer-inclusive timing of one frame (dev helper): H2D of the raw uint16 stack, LCN, U-Net, D2H of the prob map."""
import importlib, sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
synth = importlib.import_module("3deecelltracker_amd.synth"); unet3d = importlib.import_module("3deecelltracker_amd.unet3d")
pre = importlib.import_module("3deecelltracker_amd.preprocess")
model = unet3d.unet3_a().set_weights_dict(synth.make_unet_weights("unet3_a", 0))
stack, _ = synth.make_stack((512, 512, 32), 600, 0)
pin = torch.from_numpy(stack).pin_memory(); out_h = torch.empty((512, 512, 32), dtype=torch.float32).pin_memory()
dev_out = torch.zeros((512, 512, 32), device="cuda")
def frame(pinned):
    d = (pin if pinned else torch.from_numpy(stack)).to("cuda", non_blocking=pinned)
    x = pre.normalize_image_device(d, 100.0)
    model.predict_volume_device(x, out=dev_out)
    if pinned: out_h.copy_(dev_out, non_blocking=True)
    else: dev_out.cpu()
for pinned in (True, False):
    for _ in range(3): frame(pinned)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): frame(pinned)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f"{'pinned' if pinned else 'pageable'} host buffers: H2D(16.8 MB u16) + LCN + U-Net + D2H(33.5 MB): {dt*1e3:.2f} ms/frame")
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): pin.to("cuda", non_blocking=True)
torch.cuda.synchronize(); print(f"H2D alone (pinned): {(time.perf_counter()-t0)*100:.3f} ms")
t0 = time.perf_counter()
for _ in range(10): out_h.copy_(dev_out, non_blocking=True)
torch.cuda.synchronize(); print(f"D2H alone (pinned): {(time.perf_counter()-t0)*100:.3f} ms")
