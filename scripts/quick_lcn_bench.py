"""Time LCN pre-processing (ct_normalize_image) on one 512x512x32 uint16 frame (dev helper, run on the GPU box)."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
synth = importlib.import_module("3deecelltracker_amd.synth")
pre = importlib.import_module("3deecelltracker_amd.preprocess")
vol = torch.from_numpy(synth.make_stack((512, 512, 32), 600, 0)[0]).cuda()
for mode in (0, 1):
    for _ in range(3): out = pre.normalize_image_device(vol, 5.0, (27, 27, 1), mode=mode)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): out = pre.normalize_image_device(vol, 5.0, (27, 27, 1), mode=mode)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(f"LCN mode {mode}: {dt*1e3:.3f} ms per 512x512x32 uint16 frame")
