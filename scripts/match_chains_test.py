import importlib, sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
synth = importlib.import_module("3deecelltracker_amd.synth"); ffn_mod = importlib.import_module("3deecelltracker_amd.ffn")
tl = importlib.import_module("3deecelltracker_amd.trackerlite"); _dev = importlib.import_module("3deecelltracker_amd._dev")
par = importlib.import_module("3deecelltracker_amd.parallel")
ffn = ffn_mod.FFN().set_weights_dict(synth.make_ffn_weights(0))
x, y = synth.make_point_pair(600, seed=100, box=(512, 512, 32))
xn, (mean, scale) = ffn_mod.normalize_points(x, return_para=True); yn = (y - mean) / scale
a, b, c = _dev.points_dev(xn), _dev.points_dev(yn), _dev.points_dev(xn)
def job(): return tl.match_device(ffn, a, b, c, 3, 3)[0]
for cus, workers in ((32, 1), (32, 2), (32, 3), (64, 3)):
    pipe = par.FramePipeline(device=0, match_cus=cus, workers=workers)
    for _ in range(workers): pipe.submit_match(job)
    pipe.drain(); torch.cuda.synchronize()
    K = 4 * workers
    t0 = time.perf_counter()
    for _ in range(K): pipe.submit_match(job)
    pipe.drain(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    print(f"match-only: {cus} CUs, {workers} chain(s): {dt*1e3:.1f} ms per match (throughput), {dt*workers*1e3:.1f} ms latency")
    pipe.close()
