"""Where a 2000-cell TrackerLite match spends its time (BASELINE config 5's match half): python scripts/probe/m2000.py  (under scripts/prof.sh for kernel stats)"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
m = lambda n: importlib.import_module("3deecelltracker_amd." + n)
synth, ffn_mod, tl, _dev = m("synth"), m("ffn"), m("trackerlite"), m("_dev")
from pathlib import Path
w = synth.load_ffn_npz(synth.TRAINED_FFN_PATH)
ffn = ffn_mod.FFN().set_weights_dict(w)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
x, y = synth.make_point_pair(n, seed=2000, box=(512, 512, 128), voxel_size=(1.0, 1.0, 1.0))
xn, (mean, scale) = ffn_mod.normalize_points(x, return_para=True)
a, b = _dev.points_dev(xn), _dev.points_dev((y - mean) / scale)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out, it = tl.match_device(ffn, a, b, a, 3, 3)
    torch.cuda.synchronize(); print(f"match_device n={n}: {(time.perf_counter()-t0)*1e3:.2f} ms, {it} iterations", flush=True)
t0 = time.perf_counter(); corr = ffn_mod.initial_matching_device(ffn, a, b, 20); torch.cuda.synchronize(); t1 = time.perf_counter()
_, _, prior = _dev.greedy_match(corr, 0.1, 0); torch.cuda.synchronize(); t2 = time.perf_counter()
r = _dev.prgls_two_ref(prior, b, a, a, 3.0, 3.0, 2000, want_posterior=False); torch.cuda.synchronize(); t3 = time.perf_counter()
print(f"ffn {1e3*(t1-t0):.2f} greedy {1e3*(t2-t1):.2f} prgls {1e3*(t3-t2):.2f} ms, {r[-1]} iterations")
r = _dev.prgls_two_ref(prior, b, a, a, 3.0, 3.0, 2000, want_posterior=True); torch.cuda.synchronize(); t4 = time.perf_counter()
print(f"prgls with posterior {1e3*(t4-t3):.2f} ms")
