"""Debug probe: where do FrameChain.run_sequence and serial run() differ (bitwise), stage by stage?  usage: python scripts/probe/seq_determinism.py [cc|watershed] [reps]"""
import importlib
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
frame = importlib.import_module("3deecelltracker_amd.frame")
method = sys.argv[1] if len(sys.argv) > 1 else "cc"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
chain = frame.FrameChain.synthetic(shape=(256, 256, 24), n_cells=150, seed=5, region_method=method)
log = []
orig_match = frame.match_device
orig_corr = chain.transformer.accurate_correction


def match_spy(ffn, s1, s2, conf, *a, **k):
    out = orig_match(ffn, s1, s2, conf, *a, **k)
    log.append(("match_in", s1.clone(), s2.clone(), conf.clone()))          # (device-side copies on the same stream: no host round trip added)
    log.append(("match_out", out[0].clone()))
    return out


def corr_spy(prob, coords, **k):
    log.append(("prob", prob.clone()))
    log.append(("corr_in", coords.real.copy()))
    out = orig_corr(prob, coords, **k)
    log.append(("corr_out", out.real.copy()))
    return out


frame.match_device = match_spy
chain.transformer.accurate_correction = corr_spy
raws = [chain.raw_t2, chain.raw_t1, chain.raw_t2, chain.raw_t1, chain.raw_t2]
seg, conf = chain.seg_real_t1, chain.confirmed_real_t1
for r in raws:
    o = chain.run(r, seg, conf); seg, conf = o["seg_real_t2"], o["coords"].real
def host(entries):
    torch.cuda.synchronize()
    return [(e[0],) + tuple(x.cpu().numpy() if hasattr(x, "is_cuda") else x for x in e[1:]) for e in entries]


want = host(log); log.clear()
for rep in range(reps):
    list(chain.run_sequence(raws, chain.seg_real_t1, chain.confirmed_real_t1))
    got = host(log); log.clear()
    bad = []
    for i, (w, g) in enumerate(zip(want, got)):
        for a, b in zip(w[1:], g[1:]):
            if a.shape != b.shape or not np.array_equal(a, b):
                d = float(np.abs(a.astype(np.float64) - b.astype(np.float64)).max()) if a.shape == b.shape else -1
                bad.append((i // 5, w[0], d, int((a != b).sum()) if a.shape == b.shape else -1))
    print(f"rep {rep}: first differences:", bad[:6])
