"""Small matches: EM iterations as one persistent launch (grid of G workgroups) against the seven launches.  Run once per setting (the switches are read once
per process):  CT_EM_PERSISTENT=0 | CT_EM_SMALL_G=1|2|4|8  python scripts/probe/em_small_g.py"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
m = lambda n: importlib.import_module("3deecelltracker_amd." + n)
synth, ffn_mod, tl, _dev, frame = m("synth"), m("ffn"), m("trackerlite"), m("_dev"), m("frame")
ffn = ffn_mod.FFN().set_weights_dict(synth.make_ffn_weights(0))          # a noise prior: many iterations
tag = f"PERSISTENT={os.environ.get('CT_EM_PERSISTENT', '-')} SMALL_G={os.environ.get('CT_EM_SMALL_G', '-')}"
res = []
for n in (30, 50, 113, 150, 250):
    x, y = synth.make_point_pair(n, seed=300 + n, box=(512, 512, 32))
    xn, (mean, scale) = ffn_mod.normalize_points(x, return_para=True)
    a, b = _dev.points_dev(xn), _dev.points_dev((y - mean) / scale)
    corr = ffn_mod.initial_matching_device(ffn, a, b, 20)
    _, _, prior = _dev.greedy_match(corr, 0.1, 0)
    for _ in range(3):
        out = _dev.prgls_two_ref(prior, b, a, a, 3.0, 3.0, 2000, want_posterior=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        out = _dev.prgls_two_ref(prior, b, a, a, 3.0, 3.0, 2000, want_posterior=False)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    it = int(out[-1])
    res.append(f"n={n}: {dt * 1e3:.3f} ms, {it} it, {dt / max(it, 1) * 1e6:.1f} us/it")
print(tag, "|", " | ".join(res))
ch = frame.FrameChain.synthetic(shape=(64, 64, 16), n_cells=50, seed=0)
raws = ([ch.raw_t2, ch.raw_t1] * 32)
for _ in range(2):
    outs = list(ch.run_sequence(raws, ch.seg_real_t1, ch.confirmed_real_t1))
torch.cuda.synchronize(); t0 = time.perf_counter()
outs = list(ch.run_sequence(raws, ch.seg_real_t1, ch.confirmed_real_t1))
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / len(raws)
print(tag, f"| cfg1 frame loop {dt * 1e3:.3f} ms per frame ({outs[-1]['prgls_iterations']} PR-GLS iterations, {outs[-1]['n_segmented']} cells)")
