import sys, time, importlib
sys.path.insert(0, '.')
import numpy as np, torch
frame = importlib.import_module('3deecelltracker_amd.frame')
def sequence(tag, **kw):
    chain = frame.FrameChain.synthetic(shape=(512, 512, 32), n_cells=600, seed=0, **kw)
    raws = [chain.raw_t2, chain.raw_t1] * 8
    list(chain.run_sequence(raws[:4], chain.seg_real_t1, chain.confirmed_real_t1))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    outs = list(chain.run_sequence(raws, chain.seg_real_t1, chain.confirmed_real_t1))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / len(raws)
    print(tag, f"{dt*1e3:.2f} ms/frame", {k: round(v, 2) for k, v in chain.sequence_spans().items()}, flush=True)
    return chain
def one(method):
    chain = frame.FrameChain.synthetic(shape=(512, 512, 32), n_cells=600, seed=0, region_method=method)
    for _ in range(7): chain.run()
    torch.cuda.synchronize()
sequence('fresh process, default args')
sequence('again')
one('watershed'); sequence('after one(watershed)')
one('cc'); sequence('after one(cc)')
c = frame.FrameChain.synthetic(shape=(512, 512, 32), n_cells=600, seed=0)
c.run(); c.run()
raws = [c.raw_t2, c.raw_t1] * 8
list(c.run_sequence(raws[:4], c.seg_real_t1, c.confirmed_real_t1))
torch.cuda.synchronize(); t0 = time.perf_counter(); list(c.run_sequence(raws, c.seg_real_t1, c.confirmed_real_t1)); torch.cuda.synchronize()
print('same chain after run()', f"{(time.perf_counter()-t0)/16*1e3:.2f} ms/frame", {k: round(v, 2) for k, v in c.sequence_spans().items()})
