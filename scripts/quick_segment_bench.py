"""Time ct_segment_centroids on the BASELINE frame (512x512x32, ~600 cells) -- run on the GPU box."""
import importlib
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

seg = importlib.import_module("3deecelltracker_amd.segment")
synth = importlib.import_module("3deecelltracker_amd.synth")

for shape, n in (((512, 512, 32), 600), ((160, 160, 16), 113)):
    prob = torch.from_numpy(synth.make_prob_map(2, shape, n)).cuda()
    for conn in (1, 3):
        for _ in range(3):
            lab, cen, siz = seg.segment_centroids_device(prob, 0.5, conn, 30)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 20
        for _ in range(reps):
            lab, cen, siz = seg.segment_centroids_device(prob, 0.5, conn, 30)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        V = prob.numel()
        print(f"{shape} conn={conn}: {dt * 1e3:.3f} ms/frame, {len(siz)} regions, fg={float((prob > 0.5).float().mean()):.3f}, "
              f"{V * 4 * 8 / dt / 1e9:.0f} GB/s (8 sweeps x 4 B/voxel)")
