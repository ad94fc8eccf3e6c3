"""The U-Net beside a synthetic streaming co-runner (scripts/probe/stream_corun.hip): S = one U-Net per frame, W = `sweeps` read-one-write-one
passes over 512 x 512 x 32 doubles per frame (the watershed's ~30), started when the previous U-Net ends -- with plain, non-temporal, and
non-temporal-store-only memory instructions.      python scripts/probe/stream_corun.py [frames]"""
import ctypes as C
import importlib
import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch  # noqa: E402

so = os.path.join(HERE, "stream_corun.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, os.path.join(HERE, "stream_corun.hip")])
P = C.CDLL(so)
P.probe_sweep.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_void_p]
frame = importlib.import_module("3deecelltracker_amd.frame")
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 40
chain = frame.FrameChain.synthetic(shape=(512, 512, 32), n_cells=600, seed=0)
t = torch
dev = chain.raw_t2.device
norm = chain.normalized(chain.raw_t2)
probs = [t.empty(tuple(norm.shape), dtype=t.float32, device=dev) for _ in range(3)]
n = norm.numel()
a = t.zeros(n, dtype=t.float64, device=dev); b = t.zeros_like(a)
S, W = t.cuda.Stream(device=dev), t.cuda.Stream(device=dev, priority=-1)


def loop(sweeps, mode, nfr):
    ev = [None] * nfr
    e0 = t.cuda.Event(enable_timing=True); e1 = t.cuda.Event(enable_timing=True)
    t.cuda.synchronize(); e0.record(S)
    for j in range(nfr):
        with t.cuda.stream(S):
            chain.unet_model.predict_volume_device(norm, chain.shrink, out=probs[j % 3])
            ev[j] = t.cuda.Event(); ev[j].record(S)
        if j >= 1 and sweeps:
            with t.cuda.stream(W):
                W.wait_event(ev[j - 1])
                for k in range(sweeps):
                    src, dst = (a, b) if k % 2 == 0 else (b, a)
                    rc = P.probe_sweep(src.data_ptr(), dst.data_ptr(), n, mode, W.cuda_stream)
                    assert rc == 0
    e1.record(S); t.cuda.synchronize()
    return e0.elapsed_time(e1) / nfr


for sweeps in (0, 15, 30):
    for mode, name in ((0, "plain"), (1, "non-temporal loads + stores"), (2, "non-temporal stores")):
        if sweeps == 0 and mode:
            continue
        loop(sweeps, mode, 6)
        r = [loop(sweeps, mode, frames) for _ in range(2)]
        # the sweeps alone
        t.cuda.synchronize(); t0 = time.perf_counter()
        for k in range(max(sweeps, 1) * 4):
            P.probe_sweep(a.data_ptr(), b.data_ptr(), n, mode, t.cuda.current_stream().cuda_stream)
        t.cuda.synchronize(); alone = (time.perf_counter() - t0) / 4 * 1e3
        print(f"{sweeps:2d} sweeps per frame ({sweeps * n * 16 / 1e9:.2f} GB), {name:28s}: U-Net stream {r[0]:.2f} {r[1]:.2f} ms per frame; the sweeps alone {alone if sweeps else 0:.2f} ms", flush=True)
