"""The driver's K = 20 window, frame by frame: host time of every yield and the stream-local spans (U-Net / regions / match + correction) of every frame.
    python scripts/probe/seq_k20.py [K]"""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np  # noqa: E402
import torch  # noqa: E402

frame = importlib.import_module("3deecelltracker_amd.frame")
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
chain = frame.FrameChain.synthetic(shape=(512, 512, 32), n_cells=600, seed=0)
raws = ([chain.raw_t2, chain.raw_t1] * K)[:K]
list(chain.run_sequence(raws[:5], chain.seg_real_t1, chain.confirmed_real_t1))
for w in range(3):
    torch.cuda.synchronize()
    ref = torch.cuda.Event(enable_timing=True); ref.record()
    t0 = time.perf_counter(); ts = []
    for out in chain.run_sequence(raws, chain.seg_real_t1, chain.confirmed_real_t1):
        ts.append((time.perf_counter() - t0) * 1e3)
    torch.cuda.synchronize(); total = (time.perf_counter() - t0) * 1e3
    spans = list(chain._seq["spans"])
    rows = {}
    for name, e0, e1 in spans:
        rows.setdefault(name, []).append((ref.elapsed_time(e0), ref.elapsed_time(e1)))
    print(f"window {w}: total {total:.2f} ms = {total / K:.3f} ms per frame")
    print("  yields at (ms):", " ".join(f"{t:.1f}" for t in ts))
    for name in ("unet", "regions", "match+correction"):
        print(f"  {name:17s} [start-end]:", " ".join(f"{a:.1f}-{b:.1f}" for a, b in rows[name][:3]), "...", " ".join(f"{a:.1f}-{b:.1f}" for a, b in rows[name][-4:]))
