"""Why is the first timed window of bench.py 1-2 % below the later ones?  Host time stamps of every frame's yield in three consecutive 64-frame
run_sequence calls after a 5-frame warm-up (fresh process): is the deficit a ramp at the start of the first window?
    python scripts/probe/seq_rampup.py"""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np  # noqa: E402
import torch  # noqa: E402

frame = importlib.import_module("3deecelltracker_amd.frame")
chain = frame.FrameChain.synthetic(shape=(512, 512, 32), n_cells=600, seed=0)
raws = [chain.raw_t2, chain.raw_t1] * 32
list(chain.run_sequence(raws[:5], chain.seg_real_t1, chain.confirmed_real_t1))
torch.cuda.synchronize()
for w in range(3):
    ts = []
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for out in chain.run_sequence(raws, chain.seg_real_t1, chain.confirmed_real_t1):
        ts.append(time.perf_counter() - t0)
    torch.cuda.synchronize(); total = time.perf_counter() - t0
    d = np.diff(np.asarray([0.0] + ts)) * 1e3
    print(f"window {w}: {total / len(raws) * 1e3:.3f} ms per frame; first yield {d[0]:.2f} ms; frames 1-8 {d[1:9].mean():.3f}, 9-16 {d[9:17].mean():.3f}, 17-32 {d[17:33].mean():.3f}, 33-63 {d[33:].mean():.3f} ms;"
          f" drain after the last yield {(total - ts[-1]) * 1e3:.2f} ms", flush=True)
