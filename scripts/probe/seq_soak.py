"""Soak of the frame loop: one run_sequence of N frames (default 3000, ~19 s): device memory before / after every 500 frames, the number of cells every
frame segments and tracks, PR-GLS iterations, and that frames with the same input and the same predecessor state give the same coordinates.
    python scripts/probe/seq_soak.py [frames]"""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np  # noqa: E402
import torch  # noqa: E402

frame = importlib.import_module("3deecelltracker_amd.frame")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
chain = frame.FrameChain.synthetic(shape=(512, 512, 32), n_cells=600, seed=0)
raws = [chain.raw_t2, chain.raw_t1] * (n // 2)
list(chain.run_sequence(raws[:6], chain.seg_real_t1, chain.confirmed_real_t1))
torch.cuda.synchronize()
m0 = torch.cuda.memory_allocated(); r0 = torch.cuda.memory_reserved()
free0, _ = torch.cuda.mem_get_info()
seen = {}
cells, iters, rounds = set(), [], []
t0 = time.perf_counter()
for i, out in enumerate(chain.run_sequence(raws, chain.seg_real_t1, chain.confirmed_real_t1)):
    c = np.asarray(out["coords"].real)
    assert np.isfinite(c).all() and c.shape == (600, 3), (i, c.shape)
    cells.add(out["n_segmented"]); iters.append(out["prgls_iterations"]); rounds.append(out["correction_rounds"])
    if i % 500 == 499:
        free, _ = torch.cuda.mem_get_info()
        print(f"frame {i + 1}: {(time.perf_counter() - t0) / (i + 1) * 1e3:.2f} ms per frame; torch allocated {torch.cuda.memory_allocated() - m0:+d} B, reserved "
              f"{torch.cuda.memory_reserved() - r0:+d} B, device free {free - free0:+d} B since the start", flush=True)
torch.cuda.synchronize()
print(f"{n} frames: cells segmented {sorted(cells)}, PR-GLS iterations min / median / max {min(iters)} / {int(np.median(iters))} / {max(iters)}, "
      f"correction rounds {min(rounds)}-{max(rounds)}; spans kept {len(chain._seq['spans'])}")
