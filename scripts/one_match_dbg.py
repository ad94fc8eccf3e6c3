import importlib, sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
synth = importlib.import_module("3deecelltracker_amd.synth"); ffn_mod = importlib.import_module("3deecelltracker_amd.ffn")
tl = importlib.import_module("3deecelltracker_amd.trackerlite"); _dev = importlib.import_module("3deecelltracker_amd._dev")
n = int(sys.argv[1]); gain, shift = float(sys.argv[2]), float(sys.argv[3])
ffn = ffn_mod.FFN().set_weights_dict(synth.make_ffn_weights(0, gain, shift))
x, y = synth.make_point_pair(n, seed=100, box=(512, 512, 32))
xn, (mean, scale) = ffn_mod.normalize_points(x, return_para=True); yn = (y - mean) / scale
a, b, c = _dev.points_dev(xn), _dev.points_dev(yn), _dev.points_dev(xn)
out, it = tl.match_device(ffn, a, b, c, 3, 3); torch.cuda.synchronize(); print("iters", it)
