import sys, time, importlib
sys.path.insert(0, "/root/repo")
import numpy as np, torch
tracker_mod = importlib.import_module("3deecelltracker_amd.tracker"); synth = importlib.import_module("3deecelltracker_amd.synth"); ffn_mod = importlib.import_module("3deecelltracker_amd.ffn")
from pathlib import Path
n, nvol = 113, 21
rng = np.random.default_rng(12)
base = rng.uniform(0, 1, (n, 3)) * np.array([168, 401, 128])
segs, trks = [], []
for _ in range(nvol):
    a = np.eye(3) + (rng.uniform(0, 1, (3, 3)) - 0.5) * 0.04
    pts = (base - base.mean(0)) @ a + base.mean(0) + rng.normal(0, 0.5, base.shape)
    segs.append(pts[rng.permutation(n)]); trks.append(pts + rng.normal(0, 0.3, base.shape))
ffn = ffn_mod.FFN().set_weights_dict(synth.load_ffn_npz(synth.TRAINED_FFN_PATH))
for chains in (1, 2, 4, 8, 12, 20):
    trk = tracker_mod.Tracker.for_matching(ffn, beta_tk=1000.0, lambda_tk=1e-5, maxiter_tk=10, ensemble=20)
    trk.ensemble_chains = chains
    trk.history.r_segmented_coordinates = segs[:-1]; trk.history.r_tracked_coordinates = trks[:-1]
    trk.cell_num_t0 = n
    trk.inject_segmentation(segs[-1])
    for _ in range(2): out = trk.predict_ensemble(nvol)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): out = trk.predict_ensemble(nvol)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f"chains {chains}: {dt*1e3:.1f} ms per ensemble prediction")
