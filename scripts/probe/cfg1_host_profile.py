"""Where does a 64 x 64 x 16 frame (BASELINE config 1: one unet3_a patch, ~50 cells) spend its 4-8 ms?  The conv stack of that frame is 0.8 ms.
Host-side profile (cProfile, cumulative) of FrameChain.run_sequence over N frames + the stream spans + a kernel-free estimate:
    python scripts/probe/cfg1_host_profile.py [frames]"""
import cProfile
import importlib
import io
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch  # noqa: E402

frame = importlib.import_module("3deecelltracker_amd.frame")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for shape, cells in (((64, 64, 16), 50), ((256, 256, 24), 150)):
    ch = frame.FrameChain.synthetic(shape=shape, n_cells=cells, seed=0)
    raws = ([ch.raw_t2, ch.raw_t1] * n)[:n]
    for _ in range(2):
        outs = list(ch.run_sequence(raws, ch.seg_real_t1, ch.confirmed_real_t1))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    outs = list(ch.run_sequence(raws, ch.seg_real_t1, ch.confirmed_real_t1))
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"== {shape} {cells} cells: {dt * 1e3:.3f} ms per frame in the loop; spans {dict((k, round(v, 3)) for k, v in ch.sequence_spans().items())}; "
          f"PR-GLS iterations {outs[-1]['prgls_iterations']}, correction rounds {outs[-1]['correction_rounds']}, cells {outs[-1]['n_segmented']}")
    pr = cProfile.Profile()
    pr.enable()
    outs = list(ch.run_sequence(raws, ch.seg_real_t1, ch.confirmed_real_t1))
    torch.cuda.synchronize()
    pr.disable()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28)
    txt = s.getvalue()
    print("\n".join(l[:200] for l in txt.splitlines()[:60]))
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(18)
    print("\n".join(l[:200] for l in s.getvalue().splitlines()[:40]))
