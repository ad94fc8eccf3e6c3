"""The frame loop alone (FrameChain.run_sequence, steady state) for rocprofv3: kernel-trace stats or one SQ counter pass of the loop.
    python scripts/probe/seqonly.py [frames]"""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch  # noqa: E402

frame = importlib.import_module("3deecelltracker_amd.frame")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
chain = frame.FrameChain.synthetic(shape=(512, 512, 32), n_cells=600, seed=0)
raws = [chain.raw_t2, chain.raw_t1] * (n // 2)
list(chain.run_sequence(raws[:4], chain.seg_real_t1, chain.confirmed_real_t1))
torch.cuda.synchronize(); t0 = time.perf_counter()
outs = list(chain.run_sequence(raws, chain.seg_real_t1, chain.confirmed_real_t1))
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / len(raws)
print(f"frame sequence {len(raws)} frames: {dt*1e3:.2f} ms per frame ({1/dt:.1f} volumes/s)", {k: round(v, 2) for k, v in chain.sequence_spans().items()})
