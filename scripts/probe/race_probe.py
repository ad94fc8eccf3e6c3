"""Concurrent batched matches on FramePipeline workers must all return the same iteration counts and coordinates."""
import importlib, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
mod = lambda n: importlib.import_module("3deecelltracker_amd." + n)
synth, ffn_mod, tl, _dev, par, pre, unet3d = mod("synth"), mod("ffn"), mod("trackerlite"), mod("_dev"), mod("parallel"), mod("preprocess"), mod("unet3d")
prio = "prio" in sys.argv; B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 2
with_unet = "unet" in sys.argv
ffn = ffn_mod.FFN().set_weights_dict(synth.make_ffn_weights(0))
rng = np.random.default_rng(100)
x = rng.uniform(0, 1, (600, 3)) * np.array([512, 512, 32 * 5.0])
y = synth.make_target(x, seed=101) if hasattr(synth, "make_target") else x * 1.02 + rng.normal(size=x.shape)
xn, (mean, scale) = ffn_mod.normalize_points(x, return_para=True) if hasattr(ffn_mod, "normalize_points") else (x, (0, 1))
yn = (y - mean) / scale
a, b = _dev.points_dev(xn, "cuda:0"), _dev.points_dev(yn, "cuda:0")
pipe = par.FramePipeline(0, match_cus=96, workers=3, priority=prio)
if "segplain" in sys.argv:                  # U-Net on an unmasked stream (full chip), match chains on their CU-masked streams
    pipe.seg_stream = torch.cuda.Stream(device="cuda:0")
if "segmasked" in sys.argv:                 # priority pipeline, but the U-Net fenced into CUs 96..255
    import ctypes as C
    _lib = mod("_lib"); h = C.c_void_p(); _lib.check(_lib.lib().ct_stream_create_cu_range(0, 96, 160, C.byref(h)), "cu"); pipe._handles.append(h)
    pipe.seg_stream = torch.cuda.ExternalStream(h.value, device="cuda:0")
ref = tl.match_device_batched(ffn, [(a, b, a)] * B, beta=3, lambda_=3); torch.cuda.synchronize()
ref_it = [it for _, it in ref]; ref_xyz = ref[0][0].cpu().numpy()
print("reference iterations", ref_it)
if with_unet:
    model = unet3d.unet3_a(device=0).set_weights_dict(synth.make_unet_weights("unet3_a", seed=0))
    stack, _ = synth.make_stack((512, 512, 32), n_cells=600, seed=0)
    raw = torch.from_numpy(stack).cuda(); prob = torch.empty((512, 512, 32), dtype=torch.float32, device="cuda")
    mmA = torch.randn(4096, 4096, device="cuda"); mmC = torch.empty_like(mmA)
    norm0 = pre.normalize_image_device(raw, 100.0, (27, 27, 1), mode=0, subtract_median=True); torch.cuda.synchronize()
ref_prior = _dev.match_front_batched(ffn._handle, [a] * B, [b] * B, 20, 0.1, 0); torch.cuda.synchronize()
ref_prior = [p.clone() for p in ref_prior]
ref_corr = ffn_mod.initial_matching_device(ffn, a, b, 20); torch.cuda.synchronize()
ref_feat = _dev.knn_features(a, 20) if hasattr(_dev, "knn_features") else None
if "front" in sys.argv:
    ref_xyz = ref_prior[0].cpu().numpy(); ref_it = [0]
if "corr" in sys.argv:
    ref_xyz = ref_corr.cpu().numpy(); ref_it = [0]
if "greedy" in sys.argv:
    ref_xyz = _dev.greedy_match(ref_corr, 0.1, 0)[2].cpu().numpy(); ref_it = [0]
if "knn" in sys.argv:
    ref_xyz = ref_feat.cpu().numpy(); ref_it = [0]
if "knnb" in sys.argv:
    ref_xyz = _dev.knn_features(b, 20).cpu().numpy(); ref_it = [0]
if "corraa" in sys.argv:
    ref_xyz = ffn_mod.initial_matching_device(ffn, a, a, 20).cpu().numpy(); ref_it = [0]
xrows = np.random.default_rng(5).normal(size=(20000, 122)).astype(np.float32)
if "predict" in sys.argv:
    ref_xyz = ffn.predict(xrows); ref_it = [0]
def job():
    if "front" in sys.argv:            # FFN + greedy only
        pr = _dev.match_front_batched(ffn._handle, [a] * B, [b] * B, 20, 0.1, 0)
        return [(p.cpu().numpy(), 0) for p in pr]
    if "corr" in sys.argv:
        return [(ffn_mod.initial_matching_device(ffn, a, b, 20).cpu().numpy(), 0) for _ in range(B)]
    if "greedy" in sys.argv:
        return [(_dev.greedy_match(ref_corr, 0.1, 0)[2].cpu().numpy(), 0) for _ in range(B)]
    if "knnb" in sys.argv:
        return [(_dev.knn_features(b, 20).cpu().numpy(), 0) for _ in range(B)]
    if "corraa" in sys.argv:
        return [(ffn_mod.initial_matching_device(ffn, a, a, 20).cpu().numpy(), 0) for _ in range(B)]
    if "knn" in sys.argv:
        return [(_dev.knn_features(a, 20).cpu().numpy(), 0) for _ in range(B)]
    if "predict" in sys.argv:
        return [(ffn.predict(xrows), 0) for _ in range(B)]
    if "prgls" in sys.argv:            # PR-GLS only, on the reference priors
        res = _dev.prgls_two_ref_batched([(ref_prior[i], b, a, a) for i in range(B)], 3.0, 3.0, 2000)
        return [(r[0].cpu().numpy(), r[3]) for r in res]
    outs = tl.match_device_batched(ffn, [(a, b, a)] * B, beta=3, lambda_=3)
    return [(o.cpu().numpy(), it) for o, it in outs]
futs = []
for k in range(24):
    if with_unet:
        with torch.cuda.stream(pipe.seg_stream):
            if "nolcn" in sys.argv:
                norm = norm0
            else:
                norm = pre.normalize_image_device(raw, 100.0, (27, 27, 1), mode=0, subtract_median=True)
            if "matmul" in sys.argv:
                for _ in range(6):
                    torch.matmul(mmA, mmA, out=mmC)
            elif "nounet" not in sys.argv:
                model.predict_volume_device(norm, out=prob)
    futs.append(pipe.submit_match(job))
bad = 0
for k, f in enumerate(futs):
    for o, it in f.result():
        same = it == ref_it[0] and np.array_equal(o, ref_xyz)
        if not same:
            bad += 1; print("job", k, "iterations", it, "max |dx|", float(np.abs(o - ref_xyz).max()))
            if ("corr" in sys.argv or "corraa" in sys.argv or "knnb" in sys.argv) and bad <= 2:
                d = np.argwhere(o != ref_xyz)
                print("   differing entries", len(d), "of", o.size, "rows", np.unique(d[:, 0])[:12], "...", "cols", np.unique(d[:, 1])[:12], "...")
                print("   row blocks (t/32)", np.unique(d[:, 0] // 32)[:20], "col blocks (r/32)", np.unique(d[:, 1] // 32)[:20])
                t0, r0 = d[0]; print("   first", t0, r0, o[t0, r0], ref_xyz[t0, r0], "nan?", np.isnan(o).any())
print("mismatching results:", bad, "of", len(futs) * B)
if with_unet and "nounet" not in sys.argv and "matmul" not in sys.argv:
    torch.cuda.synchronize()
    got = prob.clone()
    model.predict_volume_device(norm0 if "nolcn" in sys.argv else pre.normalize_image_device(raw, 100.0, (27, 27, 1), mode=0, subtract_median=True), out=prob)
    torch.cuda.synchronize()
    print("U-Net output beside the match chains == alone:", bool(torch.equal(got, prob)), float((got - prob).abs().max()))
pipe.close()
