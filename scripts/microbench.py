"""One parameterised dev tool for the component timings DESIGN.md quotes (run on the GPU box; not the contract bench).

    python scripts/microbench.py unet [X Y Z]          U-Net volume path (default 512 512 32)
    python scripts/microbench.py arch [name ...]       patch throughput of unet3_a / unet3_c / unet3_b
    python scripts/microbench.py families              A/B of the conv kernel families (CT_CONV_MATH x CT_CONV_FOLD)
    python scripts/microbench.py lcn                   ct_normalize_image on a 512x512x32 uint16 frame
    python scripts/microbench.py segment               ct_segment_centroids
    python scripts/microbench.py watershed             ct_watershed_segment vs connected components
    python scripts/microbench.py correction            ct_accurate_correction (600 cells)
    python scripts/microbench.py match [n gain shift]  FFN + greedy + PR-GLS, per-iteration time
    python scripts/microbench.py batched [n [B]]       B matches as one batched PR-GLS chain vs separate calls
    python scripts/microbench.py goodprior [n]         PR-GLS with a prior as a trained FFN gives it
    python scripts/microbench.py legacy [n ...]        legacy Tracker._predict_pos_once
    python scripts/microbench.py ensemble              20 x 600-cell matches, 1..8 chains in flight
    python scripts/microbench.py chains                match-only throughput on CU-masked streams
    python scripts/microbench.py overlap               U-Net + match on plain / prioritised / CU-partitioned streams
    python scripts/microbench.py pcie                  host-buffer-inclusive frame time
    python scripts/microbench.py hbmwrite              fill / copy ceilings for the first conv's output size
    python scripts/microbench.py frame                 the whole on-device per-frame chain, one frame at a time
"""
import ctypes as C
import importlib
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")      # the frame loop keeps 5-6 streams busy (read when HIP initialises; the package only warns)
import numpy as np  # noqa: E402
import torch  # noqa: E402


def mod(name):
    return importlib.import_module(f"3deecelltracker_amd.{name}")


def timeit(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps, out


def norm_pair(n, seed=100, box=(512, 512, 32)):
    synth, ffn_mod, _dev = mod("synth"), mod("ffn"), mod("_dev")
    x, y = synth.make_point_pair(n, seed=seed, box=box)
    xn, (mean, scale) = ffn_mod.normalize_points(x, return_para=True)
    return _dev.points_dev(xn), _dev.points_dev((y - mean) / scale)


def cmd_unet(args):
    synth, unet3d, arch = mod("synth"), mod("unet3d"), mod("arch").UNET3_A
    shape = tuple(int(v) for v in args[:3]) if len(args) >= 3 else (512, 512, 32)
    # --passthrough: the chained frame's weights (synth.make_passthrough_unet_weights) instead of the Glorot ones; --gaps: one volume at a
    # time with an idle pause in between (what a dependent frame's U-Net sees: the chip's clocks after a low-power phase)
    w = synth.make_passthrough_unet_weights("unet3_a", 0) if "--passthrough" in args else synth.make_unet_weights("unet3_a", 0)
    model = unet3d.unet3_a().set_weights_dict(w)
    vol = torch.randn(*shape, device="cuda"); out = torch.zeros_like(vol)
    if "--gaps" in args:
        import time
        ts = []
        for _ in range(12):
            torch.cuda.synchronize(); time.sleep(0.004)
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); model.predict_volume_device(vol, out=out); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        print(f"one volume at a time after a 4-ms idle pause: {np.median(ts[2:]):.2f} ms/vol (min {min(ts[2:]):.2f}, max {max(ts[2:]):.2f})")
    dt, _ = timeit(lambda: model.predict_volume_device(vol, out=out))
    _, grid = unet3d.tile_plan(shape, arch.input_shape, (24, 24, 2))
    npatch = grid[0] * grid[1] * grid[2]
    print(f"shape {shape} patches {npatch}: {dt*1e3:.2f} ms/vol  {1/dt:.2f} vol/s  {npatch*arch.flops_per_patch()/dt/1e12:.1f} TFLOP/s fp32")
    if "--layers" in args:
        import ctypes
        L = mod("_lib").lib(); h = model._handle
        L.ct_unet_set_timing(h, 1)
        for _ in range(3):
            model.predict_volume_device(vol, out=out)
        nl = L.ct_unet_num_conv_layers(h)
        ms = (ctypes.c_float * nl)(); cnt = (ctypes.c_int * nl)()
        L.ct_unet_get_timing(h, ms, cnt, nl); L.ct_unet_set_timing(h, 0)
        for i in range(nl):
            print(f"  L{i}: {ms[i]/3:.3f} ms/vol ({cnt[i]//3} launches)")


def cmd_arch(args):
    synth, unet3d, archs = mod("synth"), mod("unet3d"), mod("arch").ARCHS
    batch = {"unet3_a": 75, "unet3_c": 150, "unet3_b": 24}
    for name in (args or ["unet3_a", "unet3_c", "unet3_b"]):
        arch = archs[name]; nb = batch[name]
        model = getattr(unet3d, name)().set_weights_dict(synth.make_unet_weights(name, 0))
        x = torch.randn(nb, *arch.input_shape, device="cuda")
        dt, _ = timeit(lambda: model.predict_device(x))
        print(f"{name}: {nb} patches {dt*1e3:.2f} ms  {nb*arch.flops_per_patch()/dt/1e12:.1f} TFLOP/s  ({dt/nb*1e3:.3f} ms/patch)")


def cmd_families(args):
    for math, fold in (("f32", "0"), ("f32", "1"), ("bf16x6", "0"), ("bf16x6", "1"), ("f16x3", "0"), ("f16x3", "1")):
        env = dict(os.environ, CT_CONV_FOLD=fold, CT_CONV_MATH=math)
        out = subprocess.run([sys.executable, __file__, "unet"], env=env, capture_output=True, text=True)
        print(f"CT_CONV_MATH={math} CT_CONV_FOLD={fold}:", out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-600:])


def cmd_lcn(args):
    synth, pre = mod("synth"), mod("preprocess")
    vol = torch.from_numpy(synth.make_stack((512, 512, 32), 600, 0)[0]).cuda()
    for mode in (0, 1):
        dt, _ = timeit(lambda: pre.normalize_image_device(vol, 5.0, (27, 27, 1), mode=mode), reps=20, warm=3)
        print(f"LCN mode {mode}: {dt*1e3:.3f} ms per 512x512x32 uint16 frame")


def cmd_segment(args):
    synth, seg = mod("synth"), mod("segment")
    for shape, n in (((512, 512, 32), 600), ((160, 160, 16), 113)):
        prob = torch.from_numpy(synth.make_prob_map(2, shape, n)).cuda()
        for conn in (1, 3):
            dt, out = timeit(lambda: seg.segment_centroids_device(prob, 0.5, conn, 30), reps=20, warm=3)
            print(f"{shape} conn={conn}: {dt*1e3:.3f} ms/frame, {len(out[2])} regions, {prob.numel()*32/dt/1e9:.0f} GB/s (8 sweeps x 4 B/voxel)")


def cmd_watershed(args):
    """ct_watershed_segment (the reference's marker watershed) beside the connected-components variant, 512x512x32 / ~600 cells."""
    synth, seg = mod("synth"), mod("segment")
    stack, _ = synth.make_stack((512, 512, 32), 600, seed=0)
    prob = torch.from_numpy(np.clip((stack.astype(np.float32) - 100.0) / 600.0, 0, 1)).cuda()
    dt, out = timeit(lambda: seg.watershed_centroids_device(prob, 4.0, "min_size", 20), reps=10, warm=2)
    print(f"watershed 512x512x32: {dt*1e3:.2f} ms/frame, {len(out[1])} cells (min_size {out[3]})")
    dt, out = timeit(lambda: seg.watershed_centroids_device(prob, 4.0, "min_size", 20, want_labels=False), reps=10, warm=2)
    print(f"watershed 512x512x32, centres only: {dt*1e3:.2f} ms/frame")
    dt, out = timeit(lambda: seg.segment_centroids_device(prob, 0.5, 1, 20), reps=20, warm=3)
    print(f"connected components 512x512x32: {dt*1e3:.3f} ms/frame, {len(out[1])} regions")


def cmd_correction(args):
    synth, cit = mod("synth"), mod("coord_image_transformer")
    shape, f, n = (512, 512, 32), 5, 600
    case = synth.make_correction_case(3, shape, f, n, 10)
    vol1 = cit.Coordinates(case["vol1"], f, case["voxel_size"], "raw")
    tr = cit.CoordsToImageTransformer(shape, case["voxel_size"], f, case["subregions"], vol1)
    coords = cit.Coordinates(case["coords0"], f, case["voxel_size"], "raw")
    prob_d = torch.from_numpy(case["prob"]).cuda()
    dt, _ = timeit(lambda: tr.accurate_correction(prob_d, coords, ensemble=True), reps=10)
    print(f"accurate_correction {shape} {n} cells: {dt*1e3:.2f} ms ({tr.last_iterations} rounds)")


def cmd_match(args):
    synth, ffn_mod, tl, _dev = mod("synth"), mod("ffn"), mod("trackerlite"), mod("_dev")
    n = int(args[0]) if args else 600
    gain, shift = (float(args[1]), float(args[2])) if len(args) > 2 else (1.0, 0.0)
    ffn = ffn_mod.FFN().set_weights_dict(synth.make_ffn_weights(0, gain, shift))
    a, b = norm_pair(n)
    dt, (out, it) = timeit(lambda: tl.match_device(ffn, a, b, a, 3, 3))
    t_ffn, corr = timeit(lambda: ffn_mod.initial_matching_device(ffn, a, b, 20), warm=0)
    t_gr, _ = timeit(lambda: _dev.greedy_match(corr, 0.1, 0), warm=0)
    print(f"n={n}: match {dt*1e3:.2f} ms total, {it} PR-GLS iterations -> {(dt - t_ffn - t_gr)/max(it,1)*1e6:.1f} us/iter; "
          f"ffn {t_ffn*1e3:.2f} ms, greedy {t_gr*1e3:.2f} ms")


def cmd_batched(args):
    """B x 600-cell PR-GLS (noise prior) as one batched chain vs B separate calls."""
    synth, ffn_mod, tl, _dev = mod("synth"), mod("ffn"), mod("trackerlite"), mod("_dev")
    ffn = ffn_mod.FFN().set_weights_dict(synth.make_ffn_weights(0))
    a, b = norm_pair(int(args[0]) if args else 600)
    corr = ffn_mod.initial_matching_device(ffn, a, b, 20)
    _, _, prior = _dev.greedy_match(corr, 0.1, 0)
    t1, out = timeit(lambda: _dev.prgls_two_ref(prior, b, a, a, 3.0, 3.0, 2000, want_posterior=False), reps=2, warm=1)
    print(f"single: {t1*1e3:.1f} ms, {out[-1]} iterations -> {t1/out[-1]*1e6:.1f} us/iteration")
    for B in ([int(args[1])] if len(args) > 1 else (1, 2, 4, 8, 16)):
        tb, res = timeit(lambda: _dev.prgls_two_ref_batched([(prior, b, a, a)] * B, 3.0, 3.0, 2000), reps=2, warm=1)
        print(f"batched B={B}: {tb*1e3:.1f} ms = {tb/B*1e3:.1f} ms per match, {tb/res[0][3]*1e6:.1f} us/iteration")


def cmd_pipebatch(args):
    """Batched match jobs on the CU-masked match partition, alone and next to the U-Net stream (what bench.py's frames mode does)."""
    synth, unet3d, ffn_mod, tl, par = mod("synth"), mod("unet3d"), mod("ffn"), mod("trackerlite"), mod("parallel")
    ffn = ffn_mod.FFN().set_weights_dict(synth.make_ffn_weights(0))
    model = unet3d.unet3_a().set_weights_dict(synth.make_unet_weights("unet3_a", 0))
    vol = torch.randn(512, 512, 32, device="cuda"); out = torch.zeros_like(vol)
    a, b = norm_pair(600)
    for cus in (64, 32):
        for B in (1, 4, 8):
            pipe = par.FramePipeline(device=0, match_cus=cus, workers=1)

            def job():
                t0 = time.perf_counter()
                tl.match_device_batched(ffn, [(a, b, a)] * B, 3, 3)
                torch.cuda.current_stream().synchronize()
                return time.perf_counter() - t0
            for with_unet in (False, True):
                f = pipe.submit_match(job)
                if with_unet:
                    while not f.done():
                        with torch.cuda.stream(pipe.seg_stream):
                            model.predict_volume_device(vol, out=out)
                        pipe.seg_stream.synchronize()
                dt = f.result()
                print(f"match partition {cus} CUs, batch {B}, U-Net running: {with_unet}: {dt*1e3:.1f} ms per job = {dt/B*1e3:.1f} ms per match")
            pipe.close()


def cmd_goodprior(args):
    ffn_mod, _dev = mod("ffn"), mod("_dev")
    n = int(args[0]) if args else 600
    rng = np.random.default_rng(7)
    x = rng.uniform(0, 1, (n, 3)) * np.array([512.0, 512.0, 128.0])
    xn, _ = ffn_mod.normalize_points(x, return_para=True)
    a = np.eye(3) + (rng.uniform(0, 1, (3, 3)) - 0.5) * 0.2
    yn = xn @ a + (rng.uniform(0, 1, xn.shape) - 0.5) * 0.004
    rep = rng.choice(n, int(0.15 * n), replace=False)
    yn[rep] = rng.uniform(-0.5, 0.5, (len(rep), 3))
    perm = rng.permutation(n); yn = yn[perm]
    corr = rng.uniform(0, 0.05, (n, n)).astype(np.float32)
    keep = ~np.isin(perm, rep)
    corr[np.arange(n)[keep], perm[keep]] = rng.uniform(0.7, 0.99, keep.sum()).astype(np.float32)
    corr_d = torch.from_numpy(corr).cuda()
    tg, (pairs, npairs, prior_d) = timeit(lambda: _dev.greedy_match(corr_d, 0.1, 0), reps=1, warm=1)
    print(f"greedy on sharp scores: {tg*1e3:.2f} ms, {int(npairs.item())} pairs")
    xd, yd = _dev.points_dev(xn), _dev.points_dev(yn)
    dt, out = timeit(lambda: _dev.prgls_two_ref(prior_d, yd, xd, xd, 3.0, 3.0, 2000), reps=10)
    err = np.abs(out[0].cpu().numpy()[perm[keep]] - yn[keep]).max()
    print(f"n={n}: PR-GLS {dt*1e3:.2f} ms, {out[-1]} iterations, max |moved - target| over true pairs {err:.2e} (normalised units)")


def cmd_legacy(args):
    synth, ffn_mod, tracker_mod = mod("synth"), mod("ffn"), mod("tracker")
    ffn = ffn_mod.FFN().set_weights_dict(synth.make_ffn_weights(0, 6.0, -3.0))
    for n in ([int(a) for a in args] or [113, 600]):
        box = (168, 401, 32) if n < 300 else (512, 512, 32)
        x, y = synth.make_point_pair(n, seed=n, box=box)
        trk = tracker_mod.Tracker.for_matching(ffn, beta_tk=1000.0, lambda_tk=1e-5, maxiter_tk=10)
        trk.set_volume1(x, x + 0.3); trk.inject_segmentation(y)
        dt, _ = timeit(lambda: trk._predict_pos_once(1), reps=5)
        print(f"legacy _predict_pos_once N={n}: {dt*1e3:.2f} ms (5 reps x 9 EM iterations)")


def cmd_ensemble(args):
    synth, ffn_mod, tl, par = mod("synth"), mod("ffn"), mod("trackerlite"), mod("parallel")
    ffn = ffn_mod.FFN().set_weights_dict(synth.make_ffn_weights(0))
    jobs = [norm_pair(600, seed=100 + k) for k in range(20)]

    def one(j):
        return tl.match_device(ffn, j[0], j[1], j[0], 3, 3)[0]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    outb = tl.match_device_batched(ffn, [(j[0], j[1], j[0]) for j in jobs], 3, 3); torch.cuda.synchronize()
    dtb = time.perf_counter() - t0
    t0 = time.perf_counter()
    outb = tl.match_device_batched(ffn, [(j[0], j[1], j[0]) for j in jobs], 3, 3); torch.cuda.synchronize()
    dtb = time.perf_counter() - t0
    ref = None
    for chains in (1, 4):
        par.chain_map(one, jobs[:chains], chains); torch.cuda.synchronize()
        t0 = time.perf_counter(); out = par.chain_map(one, jobs, chains); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        same = True if ref is None else all(torch.equal(a, b) for a, b in zip(ref, out))
        ref = ref or out
        print(f"ensemble of 20 matches, {chains} chain(s): {dt*1e3:.0f} ms ({dt/20*1e3:.1f} ms per match), identical to sequential: {same}")
    print(f"ensemble of 20 matches, batched PR-GLS: {dtb*1e3:.0f} ms ({dtb/20*1e3:.1f} ms per match), identical to sequential: "
          f"{all(float((a - b[0]).abs().max()) <= 1e-9 for a, b in zip(ref, outb))} (tracked set to 1e-9: moved once with the summed coefficients)")


def cmd_chains(args):
    synth, ffn_mod, tl, par = mod("synth"), mod("ffn"), mod("trackerlite"), mod("parallel")
    ffn = ffn_mod.FFN().set_weights_dict(synth.make_ffn_weights(0))
    a, b = norm_pair(600)

    def job():
        return tl.match_device(ffn, a, b, a, 3, 3)[0]
    for cus, workers in ((32, 1), (32, 2), (32, 3), (64, 3), (64, 4)):
        pipe = par.FramePipeline(device=0, match_cus=cus, workers=workers)
        for _ in range(workers):
            pipe.submit_match(job)
        pipe.drain(); torch.cuda.synchronize()
        K = 4 * workers; t0 = time.perf_counter()
        for _ in range(K):
            pipe.submit_match(job)
        pipe.drain(); torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / K
        print(f"match-only: {cus} CUs, {workers} chain(s): {dt*1e3:.1f} ms per match (throughput), {dt*workers*1e3:.1f} ms latency")
        pipe.close()


def cmd_overlap(args):
    synth, unet3d, ffn_mod, tl, _lib = mod("synth"), mod("unet3d"), mod("ffn"), mod("trackerlite"), mod("_lib")
    L = _lib.lib()
    model = unet3d.unet3_a().set_weights_dict(synth.make_unet_weights("unet3_a", 0))
    ffn = ffn_mod.FFN().set_weights_dict(synth.make_ffn_weights(0))
    vol = torch.randn(512, 512, 32, device="cuda"); out = torch.zeros_like(vol)
    a, b = norm_pair(600)

    def mk(first, n):
        h = C.c_void_p(); _lib.check(L.ct_stream_create_cu_range(0, first, n, C.byref(h))); return torch.cuda.ExternalStream(h.value)

    def run(s1, s2, K=5, seg=True, match=True):
        for _ in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(K):
                if seg:
                    with torch.cuda.stream(s1):
                        model.predict_volume_device(vol, out=out)
                if match:
                    with torch.cuda.stream(s2):
                        tl.match_device(ffn, a, b, a, 3, 3)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
        return dt * 1e3
    d, e, hp = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream(priority=-1)
    print("seg alone      %.1f ms" % run(d, e, match=False))
    print("match alone    %.1f ms" % run(d, e, seg=False))
    print("both, 2 streams %.1f ms" % run(d, e))
    print("both, match hi-prio %.1f ms" % run(d, hp))
    for nm in (8, 16, 32):
        print("both, CU split %d/%d %.1f ms" % (256 - nm, nm, run(mk(nm, 256 - nm), mk(0, nm))))


def cmd_pcie(args):
    synth, unet3d, pre = mod("synth"), mod("unet3d"), mod("preprocess")
    model = unet3d.unet3_a().set_weights_dict(synth.make_unet_weights("unet3_a", 0))
    stack, _ = synth.make_stack((512, 512, 32), 600, 0)
    pin = torch.from_numpy(stack).pin_memory(); out_h = torch.empty((512, 512, 32), dtype=torch.float32).pin_memory()
    dev_out = torch.zeros((512, 512, 32), device="cuda")

    def frame(pinned):
        d = (pin if pinned else torch.from_numpy(stack)).to("cuda", non_blocking=pinned)
        x = pre.normalize_image_device(d, 100.0)
        model.predict_volume_device(x, out=dev_out)
        if pinned:
            out_h.copy_(dev_out, non_blocking=True)
        else:
            dev_out.cpu()
    for pinned in (True, False):
        dt, _ = timeit(lambda: frame(pinned), reps=10, warm=3)
        print(f"{'pinned' if pinned else 'pageable'} host buffers: H2D(16.8 MB u16) + LCN + U-Net + D2H(33.5 MB): {dt*1e3:.2f} ms/frame")
    dt, _ = timeit(lambda: pin.to("cuda", non_blocking=True), reps=10, warm=0); print(f"H2D alone (pinned): {dt*1e3:.3f} ms")
    dt, _ = timeit(lambda: out_h.copy_(dev_out, non_blocking=True), reps=10, warm=0); print(f"D2H alone (pinned): {dt*1e3:.3f} ms")


def cmd_hbmwrite(args):
    x = torch.empty(983_040_000 // 4, dtype=torch.float32, device="cuda"); y = torch.empty_like(x)
    for name, fn, nbytes in (("fill 983 MB", lambda: x.fill_(1.0), x.numel() * 4), ("copy 983 MB (read + write)", lambda: y.copy_(x), 2 * x.numel() * 4)):
        dt, _ = timeit(fn, reps=10, warm=3)
        print(f"{name}: {dt*1e3:.3f} ms = {nbytes/dt/1e12:.2f} TB/s")


def cmd_frame(args):
    """The chained per-frame pipeline (frame.FrameChain): raw stack -> LCN -> U-Net -> regions/centres -> match -> correction."""
    synth, frame = mod("synth"), mod("frame")
    chain = frame.FrameChain.synthetic(shape=(512, 512, 32), n_cells=600, seed=0, prefetch_ref="--no-prefetch" not in args)
    chain.run(); chain.enable_timing()
    dt, out = timeit(lambda: chain.run(), reps=5, warm=1)
    print(f"chained frame 512x512x32: {dt*1e3:.2f} ms  ({out['n_segmented']} cells segmented, {out['prgls_iterations']} PR-GLS iterations, "
          f"{out['correction_rounds']} correction rounds)")
    for k, v in chain.stage_times().items():
        print(f"  {k}: {v:.3f} ms")
    chain.enable_timing(False)
    raws = [chain.raw_t2, chain.raw_t1] * 6
    list(chain.run_sequence(raws[:4], chain.seg_real_t1, chain.confirmed_real_t1))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    outs = list(chain.run_sequence(raws, chain.seg_real_t1, chain.confirmed_real_t1))
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / len(raws)
    print(f"frame sequence, U-Net of frame i+1 beside the tail of frame i: {dt*1e3:.2f} ms per frame ({1/dt:.1f} volumes/s), "
          f"{[o['n_segmented'] for o in outs[:4]]} cells, {[o['prgls_iterations'] for o in outs[:4]]} PR-GLS iterations")
    print("   spans on their own streams (ms):", {k: round(v, 2) for k, v in chain.sequence_spans().items()})


def cmd_trace(args):
    """Per-workgroup phase timeline of one conv layer (needs a -DCT_TRACE build: scripts/build_variants.sh trace "-DCT_TRACE";
    CTAMD_LIB=.../libctamd_trace.so python scripts/microbench.py trace <layer> ...).  Marks are s_memtime stamps (shader cycles; the
    counters of different XCDs are not aligned, so only differences inside one workgroup are used): 0 kernel entry, 1 first tile's
    loads issued, 2 first barrier passed, 3 tile staged (loads arrived, split, written), 4 last MFMA issued, 5 epilogue stores
    issued, 6 stores complete."""
    import ctypes
    synth, unet3d = mod("synth"), mod("unet3d")
    L = mod("_lib").lib()
    model = unet3d.unet3_a().set_weights_dict(synth.make_unet_weights("unet3_a", 0))
    vol = torch.randn(512, 512, 32, device="cuda"); out = torch.zeros_like(vol)
    for _ in range(2):
        model.predict_volume_device(vol, out=out)
    torch.cuda.synchronize()
    for layer in [int(a) for a in args] or [1, 12, 13]:
        nwg = 75 * 40 * 20 * 4
        buf = torch.zeros(nwg * 8, dtype=torch.int64, device="cuda")
        ctypes.c_void_p.in_dll(L, "ct_trace_buf").value = buf.data_ptr()
        ctypes.c_int.in_dll(L, "ct_trace_layer").value = layer
        model.predict_volume_device(vol, out=out); torch.cuda.synchronize()
        ctypes.c_int.in_dll(L, "ct_trace_layer").value = -1
        t = buf.cpu().numpy().reshape(-1, 8)
        t = t[t[:, 0] != 0]
        print(f"L{layer}: {len(t)} workgroups; cycles between marks")
        names = ["entry->loads issued", "loads issued->barrier", "barrier->tile staged", "staged->last MFMA (all chunks)", "MFMA->stores issued",
                 "stores issued->complete"]
        for k in range(6):
            d = (t[:, k + 1] - t[:, k]).astype(np.float64)
            print(f"   {names[k]:32s} mean {d.mean():8.0f}   p10 {np.percentile(d,10):7.0f}  p50 {np.percentile(d,50):7.0f}  p90 {np.percentile(d,90):7.0f}")
        life = (t[:, 6] - t[:, 0]).astype(np.float64)
        hw = t[:, 7]; cu = ((hw >> 32) & 0xf) * 1000 + ((hw >> 13) & 0x7) * 100 + ((hw >> 8) & 0xf)       # xcc, se, cu
        print(f"   workgroup lifetime mean {life.mean():.0f} cycles; distinct (xcc, se, cu) ids seen: {len(np.unique(cu))}")


if __name__ == "__main__":
    cmds = {k[4:]: v for k, v in globals().items() if k.startswith("cmd_")}
    if len(sys.argv) < 2 or sys.argv[1] not in cmds:
        print(__doc__); sys.exit(2)
    cmds[sys.argv[1]](sys.argv[2:])
