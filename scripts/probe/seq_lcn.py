import sys, time, importlib
sys.path.insert(0, '.')
import numpy as np, torch
frame = importlib.import_module('3deecelltracker_amd.frame')
chain = frame.FrameChain.synthetic(shape=(512, 512, 32), n_cells=600, seed=0)
raws = [chain.raw_t2, chain.raw_t1] * 16
for rep in range(2):
    for flag in (True, False):
        chain.lcn_beside_unet = flag
        list(chain.run_sequence(raws[:4], chain.seg_real_t1, chain.confirmed_real_t1))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        outs = list(chain.run_sequence(raws, chain.seg_real_t1, chain.confirmed_real_t1))
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / len(raws)
        print(f"lcn_beside_unet={flag}: {dt*1e3:.2f} ms/frame ({1/dt:.1f} volumes/s)", {k: round(v, 2) for k, v in chain.sequence_spans().items()}, flush=True)
