"""The frame loop with CUs reserved for the match + correction stream (CT_SEQ_MATCH_CUS), default prior and a 48-iteration prior:
    python scripts/probe/seq_match_cus.py"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np, torch
from pathlib import Path
m = lambda n: importlib.import_module("3deecelltracker_amd." + n)
synth, frame = m("synth"), m("frame")
tr = synth.load_ffn_npz(synth.TRAINED_FFN_PATH)
rnd = synth.make_ffn_weights(0)
def mix(a):
    return {k: ({kk: ((1 - a) * vv + a * rnd[k][kk]).astype(np.float32) for kk, vv in v.items()} if isinstance(v, dict) else ((1 - a) * v + a * rnd[k]).astype(np.float32)) for k, v in tr.items()}
for tag, w in (("trained prior", tr), ("slow prior (mix 0.74)", mix(0.74))):
    for cus in (0, 8, 16, 32):
        chain = frame.FrameChain.synthetic(shape=(512, 512, 32), n_cells=600, seed=0, ffn_weights=w)
        chain.seq_match_cus = cus
        raws = [chain.raw_t2, chain.raw_t1] * 32
        list(chain.run_sequence(raws[:4], chain.seg_real_t1, chain.confirmed_real_t1))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        outs = list(chain.run_sequence(raws, chain.seg_real_t1, chain.confirmed_real_t1))
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / len(raws)
        print(f"{tag:22s} match CUs {cus:3d}: {dt*1e3:6.2f} ms per frame ({1/dt:6.1f} volumes/s), {outs[-1]['prgls_iterations']} iterations,",
              {k: round(v, 2) for k, v in chain.sequence_spans().items()}, flush=True)
