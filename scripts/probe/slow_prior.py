"""Which degraded FFN makes PR-GLS need ~40 iterations in the chained frame?  (bench.py config.slow_prior: is the frame loop still U-Net-bound then?)
    python scripts/probe/slow_prior.py"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np, torch
from pathlib import Path
m = lambda n: importlib.import_module("3deecelltracker_amd." + n)
synth, frame = m("synth"), m("frame")
tr = synth.load_ffn_npz(synth.TRAINED_FFN_PATH)
rnd = synth.make_ffn_weights(0)
def mix(a):
    out = {}
    for k, v in tr.items():
        out[k] = {kk: ((1 - a) * vv + a * rnd[k][kk]).astype(np.float32) for kk, vv in v.items()} if isinstance(v, dict) else ((1 - a) * v + a * rnd[k]).astype(np.float32)
    return out
def gain(g):
    out = dict(tr); out["w3"] = (tr["w3"] * g).astype(np.float32); out["b3"] = (tr["b3"] * g).astype(np.float32); return out
for tag, w in [("trained", tr)] + [(f"mix {a}", mix(a)) for a in (0.5, 0.7, 0.72, 0.74, 0.75, 0.76, 0.78, 0.8)]:
    chain = frame.FrameChain.synthetic(shape=(512, 512, 32), n_cells=600, seed=0, ffn_weights=w)
    out = chain.run(); out = chain.run()
    err = float(np.abs(out["coords"].real - chain.true_t2 * np.array([1.0, 1.0, 4.0])).max(axis=1).mean())
    print(f"{tag:10s}: {out['prgls_iterations']:4d} PR-GLS iterations, mean error vs true centres {err:.3f}", flush=True)
