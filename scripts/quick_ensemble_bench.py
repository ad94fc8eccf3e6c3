"""20 independent 600-cell matches (one ensemble prediction) one by one vs three chains in flight (dev helper)."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
synth = importlib.import_module("3deecelltracker_amd.synth"); ffn_mod = importlib.import_module("3deecelltracker_amd.ffn")
tl = importlib.import_module("3deecelltracker_amd.trackerlite"); _dev = importlib.import_module("3deecelltracker_amd._dev")
par = importlib.import_module("3deecelltracker_amd.parallel")
ffn = ffn_mod.FFN().set_weights_dict(synth.make_ffn_weights(0))
jobs = []
for k in range(20):
    x, y = synth.make_point_pair(600, seed=100 + k, box=(512, 512, 32))
    xn, (mean, scale) = ffn_mod.normalize_points(x, return_para=True); yn = (y - mean) / scale
    jobs.append((_dev.points_dev(xn), _dev.points_dev(yn)))
def one(j): return tl.match_device(ffn, j[0], j[1], j[0], 3, 3)[0]
ref = None
for chains in (1, 3, 4, 5, 6, 8):
    par.chain_map(one, jobs[:chains], chains); torch.cuda.synchronize()
    t0 = time.perf_counter(); out = par.chain_map(one, jobs, chains); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    same = True if ref is None else all(torch.equal(a, b) for a, b in zip(ref, out))
    if ref is None: ref = out
    print(f"ensemble of 20 matches, {chains} chain(s): {dt*1e3:.0f} ms ({dt/20*1e3:.1f} ms per match), identical to sequential: {same}")
