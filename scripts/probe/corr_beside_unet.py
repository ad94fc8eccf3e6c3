"""Debug probe: is accurate_correction reproducible while a U-Net runs on another stream?  (fixed probability map, fixed input coordinates)
usage: python scripts/probe/corr_beside_unet.py [reps] [busy: unet|lcn|none]"""
import importlib
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
frame = importlib.import_module("3deecelltracker_amd.frame")
pre = importlib.import_module("3deecelltracker_amd.preprocess")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
busy = sys.argv[2] if len(sys.argv) > 2 else "unet"
chain = frame.FrameChain.synthetic(shape=(256, 256, 24), n_cells=150, seed=5, region_method="cc")
prob = chain.probability_map(chain.raw_t2).clone()
torch.cuda.synchronize()
cap = {}
orig = chain.transformer.accurate_correction
chain.transformer.accurate_correction = lambda p, c, **k: (cap.__setitem__("c", c), orig(p, c, **k))[1]
chain.track(prob, chain.seg_real_t1, chain.confirmed_real_t1)
chain.transformer.accurate_correction = orig
coords = cap["c"]
want = orig(prob, coords, ensemble=True).real.copy(); rounds0 = chain.transformer.last_iterations
S, T = torch.cuda.Stream(), torch.cuda.Stream(priority=-1)
other = torch.empty_like(prob)
norm = pre.normalize_image_device(chain.raw_t1, 100.0)
A = torch.randn(4096, 4096, device="cuda"); B = torch.empty_like(A)
bad = 0
for r in range(reps):
    with torch.cuda.stream(S):
        for _ in range(3):
            if busy == "unet":
                chain.unet_model.predict_volume_device(norm, chain.shrink, out=other)
            elif busy == "torch":
                for _ in range(20):
                    torch.mm(A, A, out=B)
            elif busy == "fill":
                for _ in range(200):
                    other.fill_(1.0)
            elif busy == "lcn":
                for _ in range(10):
                    pre.normalize_image_device(chain.raw_t1, 100.0)
    with torch.cuda.stream(T):
        print(f"--- rep {r}", file=sys.stderr, flush=True)
        got = orig(prob, coords, ensemble=True).real
    torch.cuda.synchronize()
    if not np.array_equal(got, want):
        bad += 1
        if bad <= 4:
            d = np.abs(got - want)
            print(f"rep {r}: {int((d > 0).sum())} numbers differ, max {d.max():.4f}; rounds {chain.transformer.last_iterations} vs {rounds0}")
print(f"busy={busy}: {bad} of {reps} corrections differ")
