"""Timing of the match path alone (dev helper)."""
import importlib, sys, time
import numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
synth = importlib.import_module("3deecelltracker_amd.synth")
ffn_mod = importlib.import_module("3deecelltracker_amd.ffn")
tl = importlib.import_module("3deecelltracker_amd.trackerlite")
_dev = importlib.import_module("3deecelltracker_amd._dev")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 600
gain, shift = (float(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (1.0, 0.0)
ffn = ffn_mod.FFN().set_weights_dict(synth.make_ffn_weights(0, gain, shift))
x, y = synth.make_point_pair(n, seed=100, box=(512, 512, 32))
xn, (mean, scale) = ffn_mod.normalize_points(x, return_para=True); yn = (y - mean) / scale
a, b, c = _dev.points_dev(xn), _dev.points_dev(yn), _dev.points_dev(xn)
def sync(): torch.cuda.synchronize()
for _ in range(2): out, it = tl.match_device(ffn, a, b, c, 3, 3)
sync(); t0 = time.perf_counter()
for _ in range(5): out, it = tl.match_device(ffn, a, b, c, 3, 3)
sync(); dt = (time.perf_counter() - t0) / 5
# components
sync(); t0 = time.perf_counter()
for _ in range(5): corr = ffn_mod.initial_matching_device(ffn, a, b, 20)
sync(); t_ffn = (time.perf_counter() - t0) / 5
t0 = time.perf_counter()
for _ in range(5): pr = _dev.greedy_match(corr, 0.1, 0)
sync(); t_gr = (time.perf_counter() - t0) / 5
print(f"n={n}: match {dt*1e3:.2f} ms total, {it} PR-GLS iterations -> {(dt - t_ffn - t_gr)/max(it,1)*1e6:.1f} us/iter; ffn {t_ffn*1e3:.2f} ms, greedy {t_gr*1e3:.2f} ms")
