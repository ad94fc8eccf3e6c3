"""Reproducibility probe: every stage of a frame, run on its own stream while another stream keeps the GPU busy (a U-Net or a rocBLAS GEMM),
must give the bits it gives on an idle GPU.  usage: python scripts/probe/repro_beside_load.py [reps]
(the accurate correction failed this before its uniform-address loads became agent-scope loads: csrc/ct_correct.hip, fresh_i32)"""
import importlib
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
frame = importlib.import_module("3deecelltracker_amd.frame")
pre = importlib.import_module("3deecelltracker_amd.preprocess")
seg = importlib.import_module("3deecelltracker_amd.segment")
tl = importlib.import_module("3deecelltracker_amd.trackerlite")
_dev = importlib.import_module("3deecelltracker_amd._dev")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
chain = frame.FrameChain.synthetic(shape=(256, 256, 24), n_cells=150, seed=5)
raw = chain.raw_t2
norm = pre.normalize_image_device(raw, 100.0)
prob = chain.unet_model.predict_volume_device(norm, chain.shrink).clone()
cen = chain.regions(prob)
vs = torch.as_tensor(np.asarray(chain.transformer.voxel_size, dtype=np.float64), device=cen.device)
conf_n, para = _dev.normalize_points(_dev.points_dev(chain.confirmed_real_t1, cen.device))
s1, _ = _dev.normalize_points(_dev.points_dev(chain.seg_real_t1, cen.device), apply_para=para)
s2, _ = _dev.normalize_points(cen * vs, apply_para=para)
cap = {}
orig = chain.transformer.accurate_correction
chain.transformer.accurate_correction = lambda p, c, **k: (cap.__setitem__("c", c), orig(p, c, **k))[1]
chain.track(prob, chain.seg_real_t1, chain.confirmed_real_t1)
chain.transformer.accurate_correction = orig
coords = cap["c"]
torch.cuda.synchronize()

stages = {
    "lcn": lambda: pre.normalize_image_device(raw, 100.0),
    "unet": lambda: chain.unet_model.predict_volume_device(norm, chain.shrink).clone(),
    "watershed": lambda: torch.cat([x.reshape(-1).double() for x in seg.watershed_centroids_device(prob, 4.0, "min_size", 20)[:3]]),
    "cc": lambda: torch.cat([x.reshape(-1).double() for x in seg.segment_centroids_device(prob, 0.5, 1, 20)]),
    "match": lambda: tl.match_device(chain.ffn_model, s1, s2, conf_n, 3.0, 3.0)[0],
    "correction": lambda: torch.from_numpy(orig(prob, coords, ensemble=True).real),
}
A = torch.randn(4096, 4096, device="cuda"); B = torch.empty_like(A)
other = torch.empty_like(prob)
S, T = torch.cuda.Stream(), torch.cuda.Stream(priority=-1)
loads = {"unet": lambda: [chain.unet_model.predict_volume_device(norm, chain.shrink, out=other) for _ in range(3)],
         "gemm": lambda: [torch.mm(A, A, out=B) for _ in range(20)]}
total = 0
for name, fn in stages.items():
    want = fn().cpu().numpy().copy()
    for lname, load in loads.items():
        if name == "unet" and lname == "unet":
            continue                                             # (the model's own scratch: one volume at a time)
        bad = 0
        for r in range(reps):
            with torch.cuda.stream(S):
                load()
            with torch.cuda.stream(T):
                got = fn().cpu().numpy()
            torch.cuda.synchronize()
            bad += int(got.shape != want.shape or not np.array_equal(got, want))
        total += bad
        print(f"{name:11s} beside {lname}: {bad} of {reps} differ")
print("TOTAL", total)
