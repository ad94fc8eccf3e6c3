#!/usr/bin/env python3
"""Benchmark of the hot path: volumes/s, segment + match, 512x512x32 stack, ~600 cells.

One "step" = one frame: 3D U-Net sliding-window inference of a synthetic 512x512x32 stack (75
patches of unet3_a, reflect pad + stitch on device) AND one TrackerLite-style match of two ~600-point
sets (kNN features -> FFN all pairs -> greedy prior -> PR-GLS), inputs resident in HBM.
N GPUs: frames are independent units -> every rank processes its own frame per step (weak scaling),
followed by the all-gather of the tracked centroid sets (RCCL).

    python bench.py --gpus 1 --steps 10 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line (rank 0).  `roofline` is measured live with HIP events recorded on the launch
stream around every launch of the dominant kernel (conv3_mfma_kernel instantiation with the largest
share of time); `cpu_baseline` times the CPU oracle (torch-CPU conv3d U-Net on 32 host threads + numpy reference-formulation match) on a
bounded sample at N=1.
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

PKG = "3deecelltracker_amd"
FP32_MFMA_PEAK_TF = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_*_f32 = fp32 vector peak
BF16_MFMA_PEAK_TF = 2500.0     # MI355X_MICROARCH.md: bf16 dense MFMA peak (~2.5 PF; measured ceiling 2382)
HBM_PEAK_TBS = 8.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--shape", type=int, nargs=3, default=(512, 512, 32))
    ap.add_argument("--cells", type=int, default=600)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl == RCCL; gloo only for single-GPU dry runs)")
    ap.add_argument("--same-device", action="store_true", help="dry run: all ranks on cuda:0 (needs --backend gloo)")
    ap.add_argument("--match-cus", type=int, default=64, help="CUs reserved for the matching chains (rest: U-Net)")
    ap.add_argument("--disjoint-match-cus", action="store_true", help="give every match chain its own CU slice (measured: worse)")
    ap.add_argument("--match-workers", type=int, default=3, help="frames whose match chains are in flight concurrently")
    ap.add_argument("--no-realistic-pass", action="store_true", help="skip the informative second pass with the synthetic-trained FFN")
    ap.add_argument("--cpu-patches", type=int, default=20, help="U-Net patches timed by the CPU baseline sample")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(args.backend, rank=rank, world_size=world)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if args.same_device:
        local = 0
    torch.cuda.set_device(local)
    dev = f"cuda:{local}"

    arch = importlib.import_module(f"{PKG}.arch").UNET3_A
    synth = importlib.import_module(f"{PKG}.synth")
    unet3d = importlib.import_module(f"{PKG}.unet3d")
    ffn_mod = importlib.import_module(f"{PKG}.ffn")
    tl = importlib.import_module(f"{PKG}.trackerlite")
    _dev = importlib.import_module(f"{PKG}._dev")
    _lib = importlib.import_module(f"{PKG}._lib")
    L = _lib.lib()

    # ---- synthetic, seeded inputs (different frame per rank), resident in HBM before timing
    shape = tuple(args.shape)
    unet_w = synth.make_unet_weights("unet3_a", seed=0)
    ffn_w = synth.make_ffn_weights(seed=0)
    model = unet3d.unet3_a(device=local).set_weights_dict(unet_w)
    ffn = ffn_mod.FFN(device=local).set_weights_dict(ffn_w)
    stack, _ = synth.make_stack(shape, n_cells=args.cells, seed=rank)
    vol = torch.from_numpy(np.ascontiguousarray(synth.normalize_stack(stack)[0, :, :, :, 0])).to(dev)
    prob = torch.zeros_like(vol)
    x, y = synth.make_point_pair(args.cells, seed=100 + rank, box=shape, voxel_size=(1.0, 1.0, 4.0))
    xn, (mean, scale) = ffn_mod.normalize_points(x, return_para=True)
    yn = (y - mean) / scale
    seg1, seg2, conf = _dev.points_dev(xn, dev), _dev.points_dev(yn, dev), _dev.points_dev(xn, dev)
    centre, grid = unet3d.tile_plan(shape, arch.input_shape, (24, 24, 2))
    n_patches = grid[0] * grid[1] * grid[2]

    # Two plain streams do not interleave on this GPU (the dispatcher drains the conv kernel's workgroups first, so the
    # dependent chain of tiny matching kernels only advances between conv launches: measured step = sum, not max), and one
    # PR-GLS chain is latency-bound (~31 ms of dependent ~5 us kernels).  FramePipeline splits the CUs with masked streams
    # and lets `--match-workers` host threads each drive the match of a different frame (frames are independent units).
    par = importlib.import_module(f"{PKG}.parallel")
    pipe = par.FramePipeline(device=local, match_cus=args.match_cus, workers=args.match_workers, disjoint=args.disjoint_match_cus)
    s_seg = pipe.seg_stream
    n_cu, k_match = pipe.n_cu, pipe.match_cus
    gather_buf = [torch.empty((args.cells, 3), dtype=torch.float64, device=dev) for _ in range(world)] if world > 1 else None
    iters_log = []
    pending = []

    active = {"ffn": ffn}

    def match_job():
        tracked, iters = tl.match_device(active["ffn"], seg1, seg2, conf, beta=3, lambda_=3)
        iters_log.append(iters)
        return tracked

    def collect(fut):
        tracked = fut.result()
        if world > 1:
            dist.all_gather(gather_buf, tracked)           # "gather of centroid sets" (14 KB / rank), main thread only
        return tracked

    def step():
        with torch.cuda.stream(s_seg):
            model.predict_volume_device(vol, out=prob)
        pending.append(pipe.submit_match(match_job))
        while len(pending) > args.match_workers:
            collect(pending.pop(0))

    def finish():
        while pending:
            collect(pending.pop(0))

    def sync_all():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    finish()
    sync_all()
    L.ct_unet_set_timing(model._handle, 1)
    iters_log.clear()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    finish()                                   # every frame's match (and gather) has completed
    sync_all()
    dt = time.perf_counter() - t0
    L.ct_unet_set_timing(model._handle, 0)
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # ---- informative second pass (never the headline value): the same pipeline with an FFN that discriminates -- the small
    # model trained on synthetic pairs by tests/golden/train_synthetic_ffn.py -- so that PR-GLS converges in a handful of
    # iterations as it does with the reference's trained weights instead of the 364 a random-init FFN's noise prior needs
    realistic = None
    trained_path = ROOT / "tests" / "golden" / "ffn_synthetic_trained.npz"
    if trained_path.exists() and not args.no_realistic_pass:
        iters_main = list(iters_log)
        active["ffn"] = ffn_mod.FFN(device=local).set_weights_dict(synth.load_ffn_npz(trained_path))
        for _ in range(args.warmup):
            step()
        finish(); sync_all(); iters_log.clear()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        finish(); sync_all()
        dt2 = time.perf_counter() - t1
        if world > 1:
            tmax = torch.tensor([dt2], dtype=torch.float64, device=dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt2 = float(tmax.item())
        realistic = {"volumes_per_s": round(world * args.steps / dt2, 3), "ms_per_step": round(dt2 / args.steps * 1e3, 3),
                     "prgls_iterations": int(np.median(iters_log)) if iters_log else None,
                     "ffn": "tests/golden/ffn_synthetic_trained.npz (synthetic-pair training, 84 % of the true pairs found at 600 cells)",
                     "note": "same partition and inputs as the headline run; only the FFN weights differ"}
        active["ffn"] = ffn
        iters_log[:] = iters_main

    # ---- roofline of the dominant kernel from the live HIP-event log
    nl = L.ct_unet_num_conv_layers(model._handle)
    ms = (C.c_float * nl)(); cnt = (C.c_int * nl)()
    _lib.check(L.ct_unet_get_timing(model._handle, ms, cnt, nl), "ct_unet_get_timing")
    by_kernel = {}
    layers = []
    for i in range(nl):
        cin, cout, nt = C.c_int(), C.c_int(), C.c_int(); d = (C.c_int * 3)()
        L.ct_unet_layer_info(model._handle, i, C.byref(cin), C.byref(cout), d, C.byref(nt))
        flops = 2.0 * d[0] * d[1] * d[2] * 27 * cin.value * cout.value * n_patches       # per launch (one volume)
        abytes = 4.0 * d[0] * d[1] * d[2] * (cin.value + cout.value) * n_patches
        code = nt.value
        bf = abs(code) >= 1000
        if bf:
            code = code - 1000 if code > 0 else code + 1000
        if code == 0:
            name = "conv_first_mfma_kernel"
        elif bf:
            z8 = "true" if (code > 0 and d[2] <= 8 and os.environ.get("CT_CONV_Z8", "1") != "0") else "false"   # 8 x 8 x 8 tiles
            name = (f"conv3_bf16x6_kernel<1, true, {'true' if code == -9 else 'false'}, false>" if code < 0 else
                    f"conv3_bf16x6_kernel<{code % 100}, false, {'true' if code > 100 else 'false'}, {z8}>")
        elif code in (-8, -9):
            name = "conv3_mfma_c8_kernel" if code == -8 else "conv3_mfma_c8_fold_kernel"
        elif code > 100:
            name = f"conv3_mfma_fold_kernel<{code - 100}>"
        else:
            name = f"conv3_mfma_kernel<{code}>"
        # MFMA work actually issued: folded decoder convs run 12 instead of 27 taps on the upsampled channels (Cout = 8: 18 of 36)
        ca = max(L.ct_unet_layer_fold_channels(model._handle, i), 0)
        issued = flops * ((cin.value - ca) + ca * 12.0 / 27.0) / cin.value
        k = by_kernel.setdefault(name, {"ms": 0.0, "launches": 0, "flops": 0.0, "bytes": 0.0, "issued": 0.0, "bf": bf})
        k["ms"] += ms[i]; k["launches"] += cnt[i]; k["flops"] += flops * cnt[i]; k["bytes"] += abytes * cnt[i]
        k["issued"] += issued * cnt[i]
        layers.append({"layer": i, "cin": cin.value, "cout": cout.value, "dims": [d[0], d[1], d[2]], "kernel": name,
                       "ms": round(ms[i] / max(cnt[i], 1), 4),
                       "tflops": round(flops * cnt[i] / max(ms[i], 1e-9) / 1e9, 2),
                       "issued_tflops": round(issued * cnt[i] / max(ms[i], 1e-9) / 1e9, 2),
                       "gbps": round(abytes * cnt[i] / max(ms[i], 1e-9) / 1e6, 1)})
    dom_name = max(by_kernel, key=lambda n: by_kernel[n]["ms"])
    dom = by_kernel[dom_name]
    # flops the kernel really executes (== the reference op's 2*27*Cin*Cout per voxel unless the kernel folds upsampled taps).
    # Split-bf16 kernels execute 6 bf16 MFMA products per fp32 product and are priced against the bf16 dense peak.
    dom_fp32_equiv = dom["issued"] / (dom["ms"] * 1e-3) / 1e12 if dom["ms"] > 0 else 0.0
    achieved = dom_fp32_equiv * (6.0 if dom["bf"] else 1.0)
    peak_tf = BF16_MFMA_PEAK_TF if dom["bf"] else FP32_MFMA_PEAK_TF
    conv_ms_total = sum(k["ms"] for k in by_kernel.values()) / max(args.steps, 1)
    # HBM traffic of the dominant kernel: rocprofv3 PMC passes cannot run inside this process; the committed
    # measurement (profiles/, scripts/prof_pmc.sh: FETCH_SIZE and WRITE_SIZE in separate passes, gfx950 x2 fetch
    # correction) is attached when it exists for this kernel
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
    roofline = {"bound": "mfma", "achieved": round(achieved, 2), "peak": peak_tf, "unit": "TFLOP/s",
                "frac": round(achieved / peak_tf, 4), "traffic": traffic, "kernel": dom_name,
                "math": "bf16x6 split (6 bf16 MFMA products per fp32 product, fp32 accumulate)" if dom["bf"] else "f32-input MFMA",
                "fp32_equivalent_tflops": round(dom_fp32_equiv, 2),
                "avg_launch_ms": round(dom["ms"] / max(dom["launches"], 1), 4), "launches": dom["launches"],
                "algorithmic_gflop_per_launch": round(dom["flops"] / max(dom["launches"], 1) / 1e9, 2),
                "executed_gflop_per_launch": round(dom["issued"] / max(dom["launches"], 1) / 1e9, 2),
                "conv_stack_ms_per_volume": round(conv_ms_total, 3),
                "conv_stack_tflops": round(n_patches * arch.flops_per_patch() / (conv_ms_total * 1e-3) / 1e12, 2) if conv_ms_total else None,
                "conv_stack_hbm_frac": round(n_patches * arch.algorithmic_bytes_per_patch() / (conv_ms_total * 1e-3) / 1e12 / HBM_PEAK_TBS, 4) if conv_ms_total else None}

    # ---- CPU baseline: the numpy oracle (reference formulation) on the host cores, bounded sample
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import match_ref as mr
        from oracle import unet_ref as ur
        rng = np.random.default_rng(0)
        plan = ur.tile_plan(shape, arch.input_shape, arch.input_shape, (24, 24, 2))
        vol_h = vol.cpu().numpy()
        patches = ur.gather_patches(vol_h, plan)[:args.cpu_patches]
        # 32 threads: measured on the 256-thread GPU box 8/16/32/64 threads -> 0.122/0.114/0.085/0.197 s per patch (256: 18 s)
        cpu_threads = min(os.cpu_count() or 1, 32)
        ur.unet_forward_torch(patches[0], unet_w, arch, threads=cpu_threads)              # warm-up (thread pool, oneDNN primitives)
        tp = time.perf_counter()
        for p in patches:
            ur.unet_forward_torch(p, unet_w, arch)
        t_patch = (time.perf_counter() - tp) / len(patches)
        tm = time.perf_counter()
        corr = mr.initial_matching(lambda q: mr.ffn_forward(ffn_w, q), xn, yn, 20)
        prior, _ = mr.simple_match(corr)
        _, _, it_cpu = mr.prgls_with_two_ref(prior, yn, xn, xn, beta=3, lambda_=3, return_iters=True)
        t_match = time.perf_counter() - tm
        t_vol = n_patches * t_patch + t_match
        cpu = {"value": round(1.0 / t_vol, 6), "unit": "volumes/s", "cores": cpu_threads, "kind": "port",
               "sample": f"{len(patches)} of {n_patches} unet3_a patches ({t_patch:.3f} s/patch, fp32 torch-CPU conv3d on {cpu_threads} host threads, the fastest count measured) "
                         f"+ one full {args.cells}-cell match ({t_match:.2f} s, {it_cpu} PR-GLS iterations); "
                         f"volume time extrapolated as {n_patches} x patch + match"}

    if rank == 0:
        value = world * args.steps / dt
        out = {
            "metric": "volumes/s segment+match, 512x512x32 stack ~600 cells",
            "value": round(value, 3), "unit": "volumes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32 (U-Net convs: fp32 in/out, exact 3-way bf16 split on the matrix cores, fp32 accumulate; FFN f32) / f64 (PR-GLS)", "data": "synthetic",
            "config": {"workload": f"{shape[0]}x{shape[1]}x{shape[2]} synthetic stack, unet3_a sliding window "
                                   f"({n_patches} patches, shrink 24,24,2) + {args.cells}-cell TrackerLite match "
                                   f"(FFN all pairs, greedy prior, PR-GLS beta=lambda=3), seeded random-init weights",
                       "patches_per_volume": n_patches, "cells": args.cells,
                       "prgls_iterations": int(np.median(iters_log)) if iters_log else None,
                       "with_discriminating_ffn": realistic,
                       "cu_partition": {"unet": n_cu - k_match, "match": k_match}, "match_chains_in_flight": args.match_workers,
                       "parallelism": f"frames sharded, {world} rank(s), all-gather of tracked centroids"},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "layers": layers,
        }
        print(json.dumps(out))
    pipe.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
