import sys, importlib, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
tl = importlib.import_module("3deecelltracker_amd.trackerlite"); synth = importlib.import_module("3deecelltracker_amd.synth"); ffn_mod = importlib.import_module("3deecelltracker_amd.ffn"); dev = importlib.import_module("3deecelltracker_amd._dev")
from pathlib import Path
ffn = ffn_mod.FFN().set_weights_dict(synth.load_ffn_npz(synth.TRAINED_FFN_PATH))
problems = []
for b, n in enumerate((2000, 1500, 1800)):
    x, y = synth.make_point_pair(n, seed=70 + b, box=(1024, 1024, 64))
    xn, (mean, scale) = ffn_mod.normalize_points(x, return_para=True); yn = (y - mean) / scale
    problems.append((dev.points_dev(xn), dev.points_dev(yn), dev.points_dev(xn[:500])))
t0 = time.perf_counter(); single = [tl.match_device(ffn, *p, beta=3, lambda_=3) for p in problems]; torch.cuda.synchronize(); t1 = time.perf_counter()
batched = tl.match_device_batched(ffn, problems, beta=3, lambda_=3); torch.cuda.synchronize(); t2 = time.perf_counter()
print("single %.1f ms, batched %.1f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3), [(int(i), int(j), bool(float((a - b).abs().max()) <= 1e-9)) for (a, i), (b, j) in zip(single, batched)])
