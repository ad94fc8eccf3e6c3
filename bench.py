#!/usr/bin/env python3
"""Benchmark of the hot path: volumes/s, segment + match, 512x512x32 stack, ~600 cells.

One "step" = one frame: LCN pre-processing + 3D U-Net sliding-window inference of a synthetic 512x512x32 uint16 stack (75
patches of unet3_a, reflect pad + stitch on device) AND one TrackerLite-style match of two ~600-point sets (kNN features -> FFN
all pairs -> greedy prior -> PR-GLS), inputs resident in HBM.

    python bench.py --gpus 1 --steps 10 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

--mode frames   (default, the contract line) frames are independent units: every rank processes its own frame per step (weak
                scaling), followed by the all-gather of the tracked centroid sets (RCCL).  Intra-GPU: the U-Net on a normal-priority
                full-chip stream, the match chain(s) on high-priority streams (--partition: CU-masked streams instead);
--mode patches  BASELINE config 3: ONE frame per step, its 75 patches sharded over the N ranks, input broadcast from rank 0,
                one all_gather_into_tensor of the per-rank centre-crop slabs, match on rank 0 (strong scaling);
--mode ensemble BASELINE config 4: one ensemble prediction per step = 20 source volumes x 113-cell legacy FFN + PR-GLS
                predictions sharded over the N ranks, all-gather of the predictions, device trim_mean (strong scaling).
At N > 1 the default run appends short `patches` and `ensemble` passes to config (so one SCALE run measures configs 3 and 4).

Prints ONE JSON line (rank 0).  `roofline` is measured live with HIP events recorded on the launch stream around every launch of
the dominant kernel; `cpu_baseline` times the CPU oracle (torch-CPU conv3d U-Net on 32 host threads + numpy
reference-formulation match) on a bounded sample at N=1.
"""
from __future__ import annotations

import argparse
import ctypes as C
import importlib
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
# HIP streams share 4 hardware queues by default; the frame loop keeps five to six streams busy (U-Net, watershed + its helper, match +
# correction, the reference-set preparation) and two of them landing on one queue serialises them (measured: the frame sequence 6.8 ms or
# 7.8-9.3 ms per frame depending on the order in which the process created its streams; 6.8 ms every time with 16 queues; the headline
# line does not move).  Read by the HIP runtime when it initialises, i.e. it has to be set before the first GPU call of the process.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

PKG = "3deecelltracker_amd"
FP32_MFMA_PEAK_TF = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_*_f32 = fp32 vector peak
BF16_MFMA_PEAK_TF = 2500.0     # MI355X_MICROARCH.md: bf16 dense MFMA peak (~2.5 PF; measured ceiling 2382)
# What v_mfma_f32_16x16x32_f16 sustains for 5 s on random finite operands at the package power cap (profiles/r04_mfma_ceiling.txt: 2.05 GHz,
# 1328 W; zero operands reach 2411 TFLOP/s at 2.39 GHz, which is the guide's condition).  Reported BESIDE roofline.peak, never instead of it.
F16_MFMA_SUSTAINED_TF = 1981.0
HBM_PEAK_TBS = 8.0
NOISE_LEVEL = 100.0


def mod(name):
    return importlib.import_module(f"{PKG}.{name}")


class Ctx:
    pass


def timed(ctx, step, finish, steps, warmup):
    """W untimed steps, then exactly K steps bracketed by barrier + synchronize; max over ranks."""
    import torch
    import torch.distributed as dist

    def sync_all():
        torch.cuda.synchronize(ctx.dev)
        if ctx.world > 1:
            dist.barrier()
        torch.cuda.synchronize(ctx.dev)
    for _ in range(warmup):
        step()
    finish(); sync_all()
    if ctx.on_timed_start:
        ctx.on_timed_start()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    finish(); sync_all()
    dt = time.perf_counter() - t0
    if ctx.world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=ctx.dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    return dt


def _match_batcher(ctx, args, on_results=None):
    """The matches of `--match-batch` consecutive frames go out as ONE chain of launches (trackerlite.match_device_batched) on the
    match partition; at most `--match-workers` chains are in flight.  Returns (frame_done, finish)."""
    tl = mod("trackerlite")
    pending = []
    waiting = [0]                                            # frames whose match has not been submitted yet

    def match_job(nframes):
        outs = tl.match_device_batched(ctx.active["ffn"], [(ctx.seg1, ctx.seg2, ctx.conf)] * nframes, beta=3, lambda_=3)
        ctx.iters_log.extend(it for _, it in outs)
        return [o for o, _ in outs]

    def collect(fut):
        tracked = fut.result()                               # the worker has synchronised its stream: the results are complete
        if on_results is not None:
            on_results(tracked)

    def submit(nframes):
        pending.append(ctx.pipe.submit_match(match_job, nframes))
        while len(pending) > args.match_workers:
            collect(pending.pop(0))

    def frame_done():
        waiting[0] += 1
        if waiting[0] >= args.match_batch:
            submit(waiting[0]); waiting[0] = 0

    def finish():
        if waiting[0]:
            submit(waiting[0]); waiting[0] = 0
        while pending:
            collect(pending.pop(0))
    return frame_done, finish


def make_frames_mode(ctx, args):
    """frames sharded: each rank its own frame (LCN -> U-Net on the big CU partition, match chains on the small one)."""
    import torch
    import torch.distributed as dist
    pre = mod("preprocess")
    comm = torch.cuda.Stream(device=ctx.dev) if ctx.world > 1 else None
    gatherer = mod("parallel").TrackedSetGather(comm)        # one all_gather_into_tensor per match batch (up to 32 frames x 14 KB per rank)

    def gather(tracked):
        if ctx.world > 1:
            gatherer(tracked)
            ctx.gathered_sets = gatherer.gathered

    frame_done, finish_matches = _match_batcher(ctx, args, gather)

    def step():
        if args.lcn_stream == "match" or (args.lcn_stream == "auto" and not ctx.pipe.match_cus):
            # light matches (priority-stream pipeline): the LCN (HBM-bound, 0.3 ms) runs beside the match chains, one frame ahead of
            # the U-Net that consumes it (158 -> 164 volumes/s).  With the 364-iteration matches of the headline run both halves of
            # the 160/96 partition are full and moving the LCN over costs throughput (118 -> 110), so it stays on the U-Net stream.
            prep, seg = ctx.pipe.prep_stream, ctx.pipe.seg_stream
            with torch.cuda.stream(prep):
                norm = pre.normalize_image_device(ctx.raw, NOISE_LEVEL, (27, 27, 1), mode=0, subtract_median=True)
                ready = prep.record_event()
            with torch.cuda.stream(seg):
                seg.wait_event(ready)
                ctx.model.predict_volume_device(norm, out=ctx.prob)
                norm.record_stream(seg)
        else:
            with torch.cuda.stream(ctx.pipe.seg_stream):
                norm = pre.normalize_image_device(ctx.raw, NOISE_LEVEL, (27, 27, 1), mode=0, subtract_median=True)
                ctx.model.predict_volume_device(norm, out=ctx.prob)
        frame_done()

    def finish():
        finish_matches()
        if ctx.pipe._prep_stream is not None:
            ctx.pipe._prep_stream.synchronize()
        if comm is not None:
            comm.synchronize()
    return step, finish


def make_patches_mode(ctx, args):
    """config 3: one frame per step, patches sharded over the ranks (parallel.predict_volume_sharded), matches on rank 0 (batched
    like the frames mode's)."""
    import torch
    import torch.distributed as dist
    pre = mod("preprocess"); par = mod("parallel")
    comm = torch.cuda.Stream(device=ctx.dev) if ctx.world > 1 else None
    frame_done, finish_matches = _match_batcher(ctx, args)

    def step():
        with torch.cuda.stream(ctx.pipe.seg_stream):
            if ctx.world > 1:
                dist.broadcast(ctx.raw.view(torch.uint8), src=0)   # the frame's raw uint16 stack (16.8 MB) reaches every rank
            norm = pre.normalize_image_device(ctx.raw, NOISE_LEVEL, (27, 27, 1), mode=0, subtract_median=True)
            ctx.prob = par.predict_volume_sharded(ctx.model, norm, src=None, comm_stream=comm)
        if ctx.rank == 0:
            frame_done()

    def finish():
        if ctx.rank == 0:
            finish_matches()
    return step, finish


def make_ensemble_mode(ctx, args):
    """config 4: 20 source volumes x 113 cells (worm4), legacy FFN + PR-GLS predictions sharded over the ranks + trim_mean."""
    synth, tracker_mod = mod("synth"), mod("tracker")
    n, nvol = 113, 21
    rng = np.random.default_rng(12)
    base = rng.uniform(0, 1, (n, 3)) * np.array([168, 401, 128])
    segs, trks = [], []
    for _ in range(nvol):
        a = np.eye(3) + (rng.uniform(0, 1, (3, 3)) - 0.5) * 0.04
        pts = (base - base.mean(0)) @ a + base.mean(0) + rng.normal(0, 0.5, base.shape)
        segs.append(pts[rng.permutation(n)]); trks.append(pts + rng.normal(0, 0.3, base.shape))
    trk = tracker_mod.Tracker.for_matching(ctx.ffn_trained or ctx.ffn, beta_tk=1000.0, lambda_tk=1e-5, maxiter_tk=10, ensemble=20)
    trk.history.r_segmented_coordinates = segs[:-1]; trk.history.r_tracked_coordinates = trks[:-1]
    trk.cell_num_t0 = n
    trk.inject_segmentation(segs[-1])

    def step():
        ctx.ensemble_out = trk.predict_ensemble(nvol)        # get_reference_vols(20, 21) = volumes 1..20

    return step, (lambda: None)


def measure_pcie(ctx):
    """Host-buffer-inclusive frame: H2D of the raw uint16 stack (pinned), LCN, U-Net, D2H of the fp32 probability map."""
    import torch
    pre = mod("preprocess")
    pin = ctx.raw.cpu().pin_memory()
    out_h = torch.empty(tuple(ctx.raw.shape), dtype=torch.float32).pin_memory()

    def frame():
        d = pin.to(ctx.dev, non_blocking=True)
        norm = pre.normalize_image_device(d, NOISE_LEVEL, (27, 27, 1), mode=0, subtract_median=True)
        ctx.model.predict_volume_device(norm, out=ctx.prob)
        out_h.copy_(ctx.prob, non_blocking=True)
    for _ in range(3):
        frame()
    torch.cuda.synchronize(ctx.dev); t0 = time.perf_counter()
    for _ in range(10):
        frame()
    torch.cuda.synchronize(ctx.dev)
    dt = (time.perf_counter() - t0) / 10
    return {"segment_ms_per_frame": round(dt * 1e3, 3), "segment_volumes_per_s": round(1.0 / dt, 2),
            "what": "pinned H2D of the raw uint16 stack + LCN + U-Net + pinned D2H of the fp32 probability map, full chip, one frame at a time"}


def measure_chained(ctx, args):
    """The chained per-frame pipeline (frame.FrameChain): the match consumes the centroids of the probability map just
    produced and the correction runs on that map -- one frame at a time on one stream.  The region step is the reference's own marker
    watershed (Tracker._segment's default, tracker.py:636-648, 671-684); the same chain with threshold + connected components
    (the cheap variant) is reported beside it."""
    import torch
    frame = mod("frame")

    def one(region_method):
        chain = frame.FrameChain.synthetic(shape=tuple(args.shape), n_cells=args.cells, seed=0, device=ctx.local, region_method=region_method)
        for _ in range(2):
            out = chain.run()
        chain.enable_timing()
        torch.cuda.synchronize(ctx.dev); t0 = time.perf_counter()
        K = 12                                   # (5 until the end of round 4: one slow frame in five moved the figure by 8 %)
        for _ in range(K):
            out = chain.run()
        torch.cuda.synchronize(ctx.dev)
        dt = (time.perf_counter() - t0) / K
        err = float(np.abs(out["coords"].real - chain.true_t2 * np.array([1.0, 1.0, 4.0])).max(axis=1).mean())
        return {"volumes_per_s": round(1.0 / dt, 2), "ms_per_frame": round(dt * 1e3, 3),
                "stage_ms": {k: round(v, 3) for k, v in chain.stage_times().items()},
                "cells_segmented": out["n_segmented"], "prgls_iterations": out["prgls_iterations"],
                "correction_rounds": out["correction_rounds"], "mean_abs_error_vs_true_centres": round(err, 3)}

    def sequence():
        """FrameChain.run_sequence: the same frames as a software pipeline -- frame i is matched against frame i-1's segmentation and moves
        frame i-1's corrected cells (every dependency of the reference's loop over volumes kept), the U-Net of frame i+2 and the watershed of
        frame i+1 run beside the match + correction of frame i.  Values identical to serial frames (tests/test_gpu_bench.py)."""
        chain = frame.FrameChain.synthetic(shape=tuple(args.shape), n_cells=args.cells, seed=0, device=ctx.local)
        raws = [chain.raw_t2, chain.raw_t1] * 16               # 32 frames (16 until the end of round 4: fill and drain, ~8 ms, are inside the timed region)
        list(chain.run_sequence(raws[:4], chain.seg_real_t1, chain.confirmed_real_t1))
        torch.cuda.synchronize(ctx.dev); t0 = time.perf_counter()
        outs = list(chain.run_sequence(raws, chain.seg_real_t1, chain.confirmed_real_t1))
        torch.cuda.synchronize(ctx.dev)
        dt = (time.perf_counter() - t0) / len(raws)
        return {"volumes_per_s": round(1.0 / dt, 2), "ms_per_frame": round(dt * 1e3, 3), "frames": len(raws),
                "stream_spans_ms": {k: round(v, 3) for k, v in chain.sequence_spans().items()},
                "cells_segmented": [o["n_segmented"] for o in outs[:2]], "prgls_iterations": [o["prgls_iterations"] for o in outs[:2]],
                "what": "every frame: LCN -> U-Net -> marker watershed -> match against the PREVIOUS frame's segmentation -> correction of the previous "
                        "frame's corrected cells; three HIP streams, one host thread (fill and drain of the pipeline inside the timed region)"}

    res = one("watershed")
    cc = one("cc")
    try:                                       # (an informative pass: a failure here must not cost the headline line)
        res["frame_sequence"] = sequence()
    except Exception as e:  # noqa: BLE001
        res["frame_sequence"] = {"error": repr(e)[:300]}
    res["region_step"] = "ct_watershed_segment (the reference's marker watershed, bit-identical to watershed.py on scikit-image: tests/test_watershed_pin.py)"
    res["with_connected_components_instead"] = {k: cc[k] for k in ("volumes_per_s", "ms_per_frame", "stage_ms", "cells_segmented")}
    res["what"] = ("raw stack -> LCN -> U-Net (pass-through weights) -> marker watershed -> centres -> FFN (synthetic-trained) + greedy + PR-GLS -> "
                   "accurate correction on the same probability map; one frame at a time (what depends only on frame t1 -- its Gram matrix's "
                   "low-rank factor -- is prepared on a second stream beside the U-Net), full chip")
    return res


def roofline_from_timing(ctx, args, n_patches, steps):
    L = ctx.L; model = ctx.model; arch = ctx.arch
    nl = L.ct_unet_num_conv_layers(model._handle)
    ms = (C.c_float * nl)(); cnt = (C.c_int * nl)()
    ctx._lib.check(L.ct_unet_get_timing(model._handle, ms, cnt, nl), "ct_unet_get_timing")
    by_kernel = {}
    layers = []
    carry = None
    for i in range(nl):
        cin, cout, nt = C.c_int(), C.c_int(), C.c_int(); d = (C.c_int * 3)()
        L.ct_unet_layer_info(model._handle, i, C.byref(cin), C.byref(cout), d, C.byref(nt))
        reg = (C.c_int * 4)()
        L.ct_unet_layer_region(model._handle, i, reg)            # volume path: decoder convs compute only what the centre crops depend on
        part = (reg[1] - reg[0]) * (reg[3] - reg[2]) / float(d[0] * d[1]) if d[0] * d[1] else 1.0
        flops = 2.0 * d[0] * d[1] * d[2] * 27 * cin.value * cout.value * n_patches * part       # per launch (one volume), computed part
        abytes = 4.0 * d[0] * d[1] * d[2] * (cin.value + cout.value) * n_patches * part
        if cnt[i] == 0 and i + 1 < nl and cnt[i + 1] > 0:
            # this conv ran inside the next layer's workgroups (conv_l0l1_fused_kernel): its flops join that launch, the tensor
            # between the two never reaches HBM
            carry = {"flops": flops, "cin": cin.value}
            layers.append({"layer": i, "cin": cin.value, "cout": cout.value, "dims": [d[0], d[1], d[2]], "computed_fraction": round(part, 4),
                           "kernel": "(fused into layer %d)" % (i + 1), "ms": 0.0, "fused_into_next": True})
            continue
        fused_prev = carry is not None
        if fused_prev:
            flops += carry["flops"]
            abytes = 4.0 * d[0] * d[1] * d[2] * (carry["cin"] + cout.value) * n_patches * part
        code = nt.value
        bf = abs(code) >= 1000
        f16 = abs(code) >= 2000
        if bf:
            off = 2000 if f16 else 1000
            code = code - off if code > 0 else code + off
        nprod = 3.0 if f16 else 6.0                       # matrix-pipe products executed per fp32 product
        if fused_prev:
            name = "conv_l0l1_fused_kernel"
            carry = None
        elif code == 0:
            name = "conv_first_kernel"                 # resolved below (depends on the arithmetic family of the other layers)
        elif bf:
            z8 = "true" if (code > 0 and d[2] <= 8 and os.environ.get("CT_CONV_Z8", "1") != "0") else "false"   # 8 x 8 x 8 tiles
            fl = "true" if f16 else "false"
            name = (f"conv3_split_kernel<{fl}, 1, true, {'true' if code == -9 else 'false'}, false>" if code < 0 else
                    f"conv3_split_kernel<{fl}, {code % 100}, false, {'true' if code > 100 else 'false'}, {z8}>")
        elif code in (-8, -9):
            name = "conv3_mfma_c8_kernel" if code == -8 else "conv3_mfma_c8_fold_kernel"
        elif code > 100:
            name = f"conv3_mfma_fold_kernel<{code - 100}>"
        else:
            name = f"conv3_mfma_kernel<{code}>"
        # MFMA work actually issued: folded decoder convs run 12 instead of 27 taps on the upsampled channels (Cout = 8: 18 of 36)
        ca = max(L.ct_unet_layer_fold_channels(model._handle, i), 0)
        issued = flops * ((cin.value - ca) + ca * 12.0 / 27.0) / cin.value
        k = by_kernel.setdefault(name, {"ms": 0.0, "launches": 0, "flops": 0.0, "bytes": 0.0, "issued": 0.0, "bf": bf, "nprod": nprod, "f16": f16})
        k["ms"] += ms[i]; k["launches"] += cnt[i]; k["flops"] += flops * cnt[i]; k["bytes"] += abytes * cnt[i]
        k["issued"] += issued * cnt[i]
        t_ms = ms[i] / max(cnt[i], 1)
        hbm_frac = abytes / max(t_ms * 1e-3, 1e-12) / 1e12 / HBM_PEAK_TBS
        mfma_frac = issued * (nprod if bf else 1.0) / max(t_ms * 1e-3, 1e-12) / 1e12 / (BF16_MFMA_PEAK_TF if bf else FP32_MFMA_PEAK_TF)
        layers.append({"layer": i, "cin": cin.value, "cout": cout.value, "dims": [d[0], d[1], d[2]], "computed_fraction": round(part, 4), "kernel": name,
                       "ms": round(t_ms, 4),
                       "tflops": round(flops * cnt[i] / max(ms[i], 1e-9) / 1e9, 2),
                       "issued_tflops": round(issued * cnt[i] / max(ms[i], 1e-9) / 1e9, 2),
                       "gbps": round(abytes * cnt[i] / max(ms[i], 1e-9) / 1e6, 1),
                       "hbm_frac": round(hbm_frac, 4), "mfma_frac": round(mfma_frac, 4),
                       "binding_roof": "hbm" if hbm_frac >= mfma_frac else "mfma"})
    # why a layer sits where it does: the committed SQ-counter digest of its kernel instantiation (scripts/prof_sq.sh + sq_summary.py on
    # `microbench.py unet`, full chip) merged in -- matrix-pipe busy share, other instructions issued per MFMA, the clock the CUs saw
    sq = {}
    for cand in sorted((ROOT / "profiles").glob("r*_unet_sq_summary.json"), reverse=True):
        try:
            sq = json.loads(cand.read_text())["kernels"]; sq_src = f"profiles/{cand.name}"
            break
        except Exception:
            pass
    for ly in layers:
        kk = sq.get(ly.get("kernel", ""))
        if not kk or not ly.get("ms"):
            continue
        ly["pipe_busy"] = kk.get("mfma_pipe_busy"); ly["clock_GHz"] = kk.get("effective_clock_GHz")
        if kk.get("non_mfma_insts_per_mfma") is not None:
            ly["non_mfma_insts_per_mfma"] = kk["non_mfma_insts_per_mfma"]
        ly["sq_source"] = sq_src
    # the fused first conv follows the family of the rest: fp16 matrix pipe with f16x3, f32-input MFMAs otherwise
    first_name = "conv_first_f16_kernel<5>" if any(k.get("f16") for k in by_kernel.values()) and os.environ.get("CT_FIRST_F16", "1") != "0" \
        else "conv_first_mfma_kernel"
    if "conv_first_kernel" in by_kernel:
        by_kernel[first_name] = by_kernel.pop("conv_first_kernel")
        layers[0]["kernel"] = first_name
    dom_name = max(by_kernel, key=lambda n: by_kernel[n]["ms"])
    dom = by_kernel[dom_name]
    # flops the kernel really executes (== the reference op's 2*27*Cin*Cout per voxel unless the kernel folds upsampled taps).
    # Split kernels execute 3 (fp16 hi/lo) or 6 (bf16 h/m/l) matrix-pipe products per fp32 product and are priced against the
    # 16-bit dense peak.
    dom_fp32_equiv = dom["issued"] / (dom["ms"] * 1e-3) / 1e12 if dom["ms"] > 0 else 0.0
    achieved = dom_fp32_equiv * (dom["nprod"] if dom["bf"] else 1.0)
    peak_tf = BF16_MFMA_PEAK_TF if dom["bf"] else FP32_MFMA_PEAK_TF
    conv_ms_total = sum(k["ms"] for k in by_kernel.values()) / max(steps, 1)
    # HBM traffic of the dominant kernel: rocprofv3 PMC passes cannot run inside this process; the committed measurement
    # (profiles/, scripts/prof_pmc.sh: FETCH_SIZE and WRITE_SIZE in separate passes, gfx950 x2 fetch correction) is attached
    traffic = None
    for cand in sorted((ROOT / "profiles").glob("r*_unet_hbm_traffic.json"), reverse=True):
        try:
            kk = json.loads(cand.read_text())["kernels"].get(dom_name)
            if kk:
                traffic = {"hbm_bytes_per_launch": round(kk["hbm_bytes_per_launch"]), "source": f"profiles/{cand.name}",
                           "algorithmic_bytes_per_launch": round(dom["bytes"] / max(dom["launches"], 1))}
                break
        except Exception:
            pass
    first = layers[0] if not layers[0].get("fused_into_next") else layers[1]
    hbm_contract = round(n_patches * arch.algorithmic_bytes_per_patch() / (conv_ms_total * 1e-3) / 1e12 / HBM_PEAK_TBS, 4) if conv_ms_total else None
    roofline = {"bound": "mfma", "achieved": round(achieved, 2), "peak": peak_tf, "unit": "TFLOP/s",
                "frac": round(achieved / peak_tf, 4),
                # the matrix pipe's rate on realistic operands at the power cap, and the kernel against THAT (context; `peak` stays the datasheet figure)
                "sustained_peak": F16_MFMA_SUSTAINED_TF if dom.get("f16") else None,
                "frac_of_sustained_peak": round(achieved / F16_MFMA_SUSTAINED_TF, 4) if dom.get("f16") else None,
                # SURVEY 8(d)'s contract figure: the reference operator's minimum activation traffic (285.1 MB per unet3_a patch) x patches over the
                # conv stack's time, against the 8 TB/s HBM peak -- the number the north-star's ">= 60 % HBM roofline" is read on
                "hbm_contract_frac": hbm_contract,
                # the reference operator's flops (2 * 27 * Cin * Cout per computed voxel; one product per fp32 product) over the same
                # duration against the peak of the pipe the kernel runs on: `frac` is pipe utilisation, this is useful work
                "algorithmic_frac": round(dom["flops"] / (dom["ms"] * 1e-3) / 1e12 / peak_tf, 4) if dom["ms"] > 0 else None,
                # contract: HBM bytes per launch of this kernel from the PMC counters (number or null); where it comes from is traffic_detail
                "traffic": traffic["hbm_bytes_per_launch"] if traffic else None, "traffic_detail": traffic, "kernel": dom_name,
                "math": ("f16x3 split (2 fp16 components per operand, 3 MFMA products per fp32 product, per-patch power-of-two scaling, fp32 accumulate)" if dom.get("f16") else
                         "bf16x6 split (6 bf16 MFMA products per fp32 product, fp32 accumulate)") if dom["bf"] else "f32-input MFMA",
                "fp32_equivalent_tflops": round(dom_fp32_equiv, 2),
                "frac_of_cu_share": round(achieved / peak_tf * ctx.pipe.n_cu / max(ctx.pipe.n_cu - ctx.pipe.match_cus, 1), 4),
                "note": "the U-Net stream owns cu_partition.unet of the chip's CUs (the rest runs the match chains); frac is against the "
                        "FULL-chip peak, frac_of_cu_share against the peak of the CUs this kernel may use",
                "avg_launch_ms": round(dom["ms"] / max(dom["launches"], 1), 4), "launches": dom["launches"],
                "algorithmic_gflop_per_launch": round(dom["flops"] / max(dom["launches"], 1) / 1e9, 2),
                "executed_gflop_per_launch": round(dom["issued"] / max(dom["launches"], 1) / 1e9, 2),
                "conv_stack_ms_per_volume": round(conv_ms_total, 3),
                "conv_stack_tflops": round(n_patches * arch.flops_per_patch() / (conv_ms_total * 1e-3) / 1e12, 2) if conv_ms_total else None,
                "conv_stack_hbm_frac": hbm_contract,          # (the name of rounds 2-3; same number as hbm_contract_frac)
                "hbm_bound_kernel": {"kernel": first["kernel"], "layer": 0, "achieved_GBps": first["gbps"], "peak_GBps": HBM_PEAK_TBS * 1e3,
                                     "frac": first["hbm_frac"], "avg_launch_ms": first["ms"],
                                     "note": "the first conv (Cin = 1, AI 12 flop/B) is the HBM-bound instantiation; when it runs inside the second conv's "
                                             "workgroups (conv_l0l1_fused_kernel) this entry is the fused pair: it reads the 1-channel patch and writes 16 channels"}}
    return roofline, layers


def cpu_baseline(ctx, args, n_patches):
    from oracle import match_ref as mr
    from oracle import preprocess_ref as pr
    from oracle import unet_ref as ur
    arch = ctx.arch; shape = tuple(args.shape)
    plan = ur.tile_plan(shape, arch.input_shape, arch.input_shape, (24, 24, 2))
    tl0 = time.perf_counter()
    vol_h = pr.normalize_image(ctx.raw.cpu().numpy().astype(np.float64), NOISE_LEVEL).astype(np.float32)
    t_lcn = time.perf_counter() - tl0
    patches = ur.gather_patches(vol_h, plan)[:args.cpu_patches]
    # 32 threads: measured on the 256-thread GPU box 8/16/32/64 threads -> 0.122/0.114/0.085/0.197 s per patch (256: 18 s)
    cpu_threads = min(os.cpu_count() or 1, 32)
    ur.unet_forward_torch(patches[0], ctx.unet_w, arch, threads=cpu_threads)              # warm-up (thread pool, oneDNN primitives)
    tp = time.perf_counter()
    for p in patches:
        ur.unet_forward_torch(p, ctx.unet_w, arch)
    t_patch = (time.perf_counter() - tp) / len(patches)
    tm = time.perf_counter()
    corr = mr.initial_matching(lambda q: mr.ffn_forward(ctx.ffn_w, q), ctx.xn, ctx.yn, 20)
    prior, _ = mr.simple_match(corr)
    _, _, it_cpu = mr.prgls_with_two_ref(prior, ctx.yn, ctx.xn, ctx.xn, beta=3, lambda_=3, return_iters=True)
    t_match = time.perf_counter() - tm
    t_vol = t_lcn + n_patches * t_patch + t_match
    return {"value": round(1.0 / t_vol, 6), "unit": "volumes/s", "cores": cpu_threads, "kind": "port",
            "sample": f"LCN ({t_lcn:.2f} s, numpy) + {len(patches)} of {n_patches} unet3_a patches ({t_patch:.3f} s/patch, fp32 torch-CPU conv3d on "
                      f"{cpu_threads} host threads, the fastest count measured) + one full {args.cells}-cell match ({t_match:.2f} s, {it_cpu} "
                      f"PR-GLS iterations); volume time = LCN + {n_patches} x patch + match" +
                      ("" if len(patches) >= n_patches else f" (extrapolated from {len(patches)} patches)")}


def match_schedule(steps: int, partition: bool, workers: int | None, batch: int | None):
    """(match chains in flight, frames per chain).  Defaults: one chain of 32 frames on the priority-stream pipeline (143 volumes/s
    at K = 128 and at K = 20; 2 x 16: 142), three chains of 16 on the CU partition.  A short run (the driver's --steps 20) must not
    end on a queue of match batches: never more batches than chains in flight, so that every match starts while the U-Net frames
    are still running (the host enqueues far ahead of the GPU; the matches of this benchmark take given point sets, they do not
    wait for their frame's segmentation - SURVEY 8e's independent units)."""
    if workers is None:
        workers = 3 if partition else 1
    if batch is None:
        batch = 16 if partition else 32
    return workers, max(1, min(batch, -(-steps // workers)))


def relaunch_under_launcher(n: int):
    """`python bench.py --gpus N` without a launcher (the shape of the driver's N = 1 command): become
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py <same argv>`,
    one rank per GPU.  The port is one the kernel just handed out, the rendezvous address the loopback (the container's host
    name may not resolve)."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    sys.stdout.flush(); sys.stderr.flush()
    os.execv(sys.executable, cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=128)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--mode", choices=("frames", "patches", "ensemble"), default="frames")
    ap.add_argument("--shape", type=int, nargs=3, default=(512, 512, 32))
    ap.add_argument("--cells", type=int, default=600)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl == RCCL; gloo only for single-GPU dry runs)")
    ap.add_argument("--same-device", action="store_true", help="dry run: all ranks on cuda:0 (needs --backend gloo)")
    ap.add_argument("--match-cus", type=int, default=96, help="CUs reserved for the matching chains (rest: U-Net)")
    ap.add_argument("--lcn-stream", choices=("auto", "match", "seg"), default="auto",
                    help="frames mode: where the LCN of a frame runs (match: beside the match chains, one frame ahead of the U-Net; "
                         "auto: there when the pipeline has no CU partition, i.e. when the match side has slack)")
    ap.add_argument("--disjoint-match-cus", action="store_true", help="give every match chain its own CU slice (measured: worse)")
    ap.add_argument("--match-workers", type=int, default=None, help="match chains in flight concurrently (default: 1; 3 with --partition)")
    ap.add_argument("--realistic-match-cus", type=int, default=32, help="match partition of the informative pass with the discriminating FFN")
    ap.add_argument("--partition", action="store_true",
                    help="CU-partitioned pipeline (U-Net on n_cu - match_cus CUs, match chains on --match-cus) instead of the default: U-Net on a "
                         "normal-priority full-chip stream, match chains on high-priority streams (132 vs 124 volumes/s)")
    ap.add_argument("--priority-streams", action="store_true", help="(the default now; kept for old command lines)")
    ap.add_argument("--realistic-partition", action="store_true", help="discriminating-FFN pass on a CU partition (--realistic-match-cus) instead of priority streams (116 vs 121 volumes/s)")
    ap.add_argument("--match-batch", type=int, default=None, help="frames whose matches share one chain of launches (ct_prgls_two_ref_batched); capped at ceil(steps / chains) so that a short run does not end on queued match batches")
    ap.add_argument("--no-realistic-pass", action="store_true", help="skip the informative passes (discriminating FFN, chained frame, PCIe, sharding modes)")
    ap.add_argument("--launch-check", action="store_true", help="rendezvous only: every rank joins the process group, rank 0 prints {world_size, backend}; no GPU work (CPU test of the self-launch)")
    ap.add_argument("--cpu-patches", type=int, default=75, help="U-Net patches timed by the CPU baseline sample (default: the whole 75-patch volume, ~6 s on 32 threads: nothing is extrapolated)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        relaunch_under_launcher(args.gpus)          # does not return
    args.match_workers, args.match_batch = match_schedule(args.steps, args.partition, args.match_workers, args.match_batch)

    import torch
    import torch.distributed as dist
    ctx = Ctx()
    ctx.world = int(os.environ.get("WORLD_SIZE", "1"))
    ctx.rank = int(os.environ.get("RANK", "0"))
    ctx.local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.same_device:
        ctx.local = 0
    if args.launch_check:
        if ctx.world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group(args.backend, rank=ctx.rank, world_size=ctx.world)
            seen = [None] * ctx.world
            dist.all_gather_object(seen, (ctx.rank, ctx.local))
            dist.destroy_process_group()
        else:
            seen = [(0, 0)]
        if ctx.rank == 0:
            print(json.dumps({"launch_check": True, "world_size": ctx.world, "gpus": args.gpus, "backend": args.backend if ctx.world > 1 else None,
                              "ranks": sorted(r for r, _ in seen), "local_ranks": sorted(l for _, l in seen)}))
        return
    torch.cuda.set_device(ctx.local)
    if ctx.world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(args.backend, rank=ctx.rank, world_size=ctx.world)
    if ctx.world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={ctx.world} ranks")
    ctx.dev = f"cuda:{ctx.local}"
    world, rank, dev = ctx.world, ctx.rank, ctx.dev

    ctx.arch = arch = mod("arch").UNET3_A
    synth, unet3d, ffn_mod, _dev = mod("synth"), mod("unet3d"), mod("ffn"), mod("_dev")
    ctx._lib = mod("_lib"); ctx.L = L = ctx._lib.lib()
    par = mod("parallel")

    # ---- synthetic, seeded inputs (frames mode: a different frame per rank), resident in HBM before timing
    shape = tuple(args.shape)
    ctx.unet_w = synth.make_unet_weights("unet3_a", seed=0)
    ctx.ffn_w = synth.make_ffn_weights(seed=0)
    ctx.model = unet3d.unet3_a(device=ctx.local).set_weights_dict(ctx.unet_w)
    ctx.ffn = ffn_mod.FFN(device=ctx.local).set_weights_dict(ctx.ffn_w)
    trained_path = ROOT / "tests" / "golden" / "ffn_synthetic_trained.npz"
    ctx.ffn_trained = ffn_mod.FFN(device=ctx.local).set_weights_dict(synth.load_ffn_npz(trained_path)) if trained_path.exists() else None
    frame_seed = rank if args.mode == "frames" else 0
    stack, _ = synth.make_stack(shape, n_cells=args.cells, seed=frame_seed)
    ctx.raw = torch.from_numpy(stack).to(dev)                                   # uint16, LCN runs inside the step
    ctx.prob = torch.zeros(shape, dtype=torch.float32, device=dev)
    x, y = synth.make_point_pair(args.cells, seed=100 + frame_seed, box=shape, voxel_size=(1.0, 1.0, 4.0))
    ctx.xn, (mean, scale) = ffn_mod.normalize_points(x, return_para=True)
    ctx.yn = (y - mean) / scale
    ctx.seg1, ctx.seg2, ctx.conf = _dev.points_dev(ctx.xn, dev), _dev.points_dev(ctx.yn, dev), _dev.points_dev(ctx.xn, dev)
    centre, grid = unet3d.tile_plan(shape, arch.input_shape, (24, 24, 2))
    n_patches = grid[0] * grid[1] * grid[2]

    # Two plain streams do not interleave on this GPU (the dispatcher drains the conv kernel's workgroups first, so the
    # dependent chain of tiny matching kernels only advances between conv launches: measured step = sum, not max), and one
    # PR-GLS chain is latency-bound.  FramePipeline splits the CUs with masked streams and lets `--match-workers` host threads
    # each drive the match of a different frame (frames are independent units).
    ctx.pipe = par.FramePipeline(device=ctx.local, match_cus=args.match_cus, workers=args.match_workers, disjoint=args.disjoint_match_cus, priority=not args.partition)
    ctx.iters_log = []
    ctx.gathered_sets = 0
    ctx.active = {"ffn": ctx.ffn}
    ctx.on_timed_start = None
    makers = {"frames": make_frames_mode, "patches": make_patches_mode, "ensemble": make_ensemble_mode}

    # ---- the headline pass
    step, finish = makers[args.mode](ctx, args)

    def start_timing():
        L.ct_unet_set_timing(ctx.model._handle, 1); ctx.iters_log.clear()
    ctx.on_timed_start = start_timing
    dt = timed(ctx, step, finish, args.steps, args.warmup)
    L.ct_unet_set_timing(ctx.model._handle, 0)
    ctx.on_timed_start = None
    iters_main = list(ctx.iters_log)
    roofline, layers = (None, None)
    if args.mode != "ensemble":
        roofline, layers = roofline_from_timing(ctx, args, n_patches, args.steps)
    units = world * args.steps if args.mode == "frames" else args.steps

    # ---- informative passes (never the headline value)
    extra = {}
    if not args.no_realistic_pass:
        if ctx.ffn_trained is not None and args.mode != "ensemble":
            # the same pipeline with an FFN that discriminates (trained on synthetic pairs, tests/golden/train_synthetic_ffn.py):
            # PR-GLS converges in a handful of iterations as with the reference's trained weights instead of the ~364 a
            # random-init FFN's noise prior needs
            # a match that converges in ~10 iterations needs far fewer CUs: this pass runs on its own partition
            ctx.active["ffn"] = ctx.ffn_trained
            ctx.iters_log.clear()
            headline_pipe = ctx.pipe
            ctx.pipe = par.FramePipeline(device=ctx.local, match_cus=args.realistic_match_cus, workers=args.match_workers, priority=not args.realistic_partition)
            step2, finish2 = makers[args.mode](ctx, args)
            dt2 = timed(ctx, step2, finish2, args.steps, args.warmup)
            extra["with_discriminating_ffn"] = {
                "volumes_per_s": round(units / dt2, 3), "ms_per_step": round(dt2 / args.steps * 1e3, 3),
                "prgls_iterations": int(np.median(ctx.iters_log)) if ctx.iters_log else None,
                "cu_partition": ({"unet": ctx.pipe.n_cu - ctx.pipe.match_cus, "match": ctx.pipe.match_cus} if ctx.pipe.match_cus else
                                 {"unet": ctx.pipe.n_cu, "match": "no partition: match chains on high-priority streams"}),
                "ffn": "tests/golden/ffn_synthetic_trained.npz (synthetic-pair training, 84 % of the true pairs found at 600 cells)",
                "note": "same inputs and the same pipeline as the headline run; only the FFN weights differ (10 instead of 364 PR-GLS iterations per match)"}
            ctx.pipe.close(); ctx.pipe = headline_pipe
            ctx.active["ffn"] = ctx.ffn
        if world == 1 and args.mode == "frames":
            extra["chained"] = measure_chained(ctx, args)
            extra["pcie_inclusive"] = measure_pcie(ctx)
        if world > 1 and args.mode == "frames":
            # BASELINE configs 3 and 4 inside the same launch, so that one scaling run measures them too
            k2 = max(3, min(args.steps, 10))
            for name in ("patches", "ensemble"):           # (patches: rank 0's frame is broadcast inside the step)
                try:                                       # a failure here must not cost the headline line above it
                    s2, f2 = makers[name](ctx, args)
                    d2 = timed(ctx, s2, f2, k2, 2)
                except Exception as e:                     # noqa: BLE001  (reported, not swallowed)
                    extra[f"{name}_sharded"] = {"error": f"{type(e).__name__}: {e}"[:300]}
                    break
                extra[f"{name}_sharded"] = {"per_s": round(k2 / d2, 3), "ms_per_step": round(d2 / k2 * 1e3, 3), "steps": k2, "scaling": "strong",
                                            "what": ("one 512x512x32 frame per step, 75 patches over the ranks, input broadcast + one all_gather_into_tensor of centre-crop slabs, match on rank 0"
                                                     if name == "patches" else
                                                     "one ensemble prediction per step: 20 source volumes x 113 cells (legacy FFN + PR-GLS) over the ranks, all-gather + device trim_mean")}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.mode == "frames":
        cpu = cpu_baseline(ctx, args, n_patches)

    if rank == 0:
        value = units / dt
        if args.mode == "ensemble":
            metric = "ensemble predictions/s (20 source volumes x 113 cells, legacy FFN + PR-GLS + trim_mean; BASELINE config 4)"
            workload = "20 x 113-cell legacy Tracker predictions (beta 1000, lambda 1e-5, maxiter 10, 5 repetitions) + trim_mean(0.1)"
            parallelism = f"source volumes sharded over {world} rank(s), all-gather of predictions"
        else:
            metric = "volumes/s segment+match, 512x512x32 stack ~600 cells"
            workload = (f"{shape[0]}x{shape[1]}x{shape[2]} synthetic uint16 stack, LCN (27x27x1, noise_level {NOISE_LEVEL:g}) + unet3_a sliding window "
                        f"({n_patches} patches, shrink 24,24,2) + {args.cells}-cell TrackerLite match (FFN all pairs, greedy prior, PR-GLS "
                        f"beta=lambda=3), seeded random-init weights")
            parallelism = (f"frames sharded, {world} rank(s), all-gather of tracked centroids" if args.mode == "frames" else
                           f"patches of one frame sharded over {world} rank(s), input broadcast + all-gather of centre-crop slabs, match on rank 0")
        out = {
            "metric": metric,
            "value": round(value, 3), "unit": "volumes/s" if args.mode != "ensemble" else "predictions/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak" if args.mode == "frames" else "strong",
            "vs_baseline": None,
            "dtype": "f32 (U-Net convs: fp32 in/out, fp16 hi/lo split on the matrix cores with exact power-of-two scaling, fp32 accumulate; FFN f32) / f64 (PR-GLS)",
            "data": "synthetic",
            "config": dict({"workload": workload, "mode": args.mode, "patches_per_volume": n_patches, "cells": args.cells,
                            "prgls_iterations": int(np.median(iters_main)) if iters_main else None,
                            "cu_partition": ({"unet": ctx.pipe.n_cu - ctx.pipe.match_cus, "match": ctx.pipe.match_cus} if ctx.pipe.match_cus else
                                             {"unet": ctx.pipe.n_cu, "match": "no partition: match chains on high-priority streams"}),
                            "match_chains_in_flight": args.match_workers, "frames_per_match_chain": args.match_batch,
                            "headline_excludes": ["regions->centres (ct_watershed_segment, the reference's marker watershed)", "accurate correction"] if args.mode != "ensemble" else [],
                            "headline_note": "matches take given ~600-point sets (independent units, SURVEY 8e); the dependent per-frame chain "
                                             "incl. the watershed and the correction is config.chained (one frame's latency) and "
                                             "config.chained.frame_sequence (a sequence of such frames, software-pipelined)",
                            "rccl_ranks": ({"world_size": dist.get_world_size(), "backend": dist.get_backend(),
                                            "tracked_sets_gathered": ctx.gathered_sets} if world > 1 else
                                           {"world_size": 1, "backend": None, "tracked_sets_gathered": 0}),
                            "parallelism": parallelism}, **extra),
            "roofline": roofline,
            "cpu_baseline": cpu,
            "layers": layers,
        }
        print(json.dumps(out))
    ctx.pipe.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
