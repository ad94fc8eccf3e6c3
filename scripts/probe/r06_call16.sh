#!/bin/bash
# round 6, call 16: 3000-frame soak of the frame loop with the round's changes (device-resident stacks), then 1000 frames with HOST stacks uploaded in the loop
export GPU_MAX_HW_QUEUES=16
mkdir -p gpurun_out
python scripts/probe/seq_soak.py 3000 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_c16_soak.txt
python - >> gpurun_out/r06_c16_soak.txt 2>&1 <<'PY'
import importlib, sys, time, numpy as np, torch
sys.path.insert(0, ".")
frame = importlib.import_module("3deecelltracker_amd.frame")
ch = frame.FrameChain.synthetic(shape=(512, 512, 32), n_cells=600, seed=0)
host = [ch.raw_t2.cpu().pin_memory(), ch.raw_t1.cpu().pin_memory()]
dev = [ch.raw_t2, ch.raw_t1]
want = [np.asarray(o["coords"].real).copy() for o in ch.run_sequence(dev * 3, ch.seg_real_t1, ch.confirmed_real_t1)]
n = 1000
torch.cuda.synchronize(); free0, _ = torch.cuda.mem_get_info(); t0 = time.perf_counter(); bad = 0
for i, o in enumerate(ch.run_sequence(host * (n // 2), ch.seg_real_t1, ch.confirmed_real_t1)):
    if i < 6 and not np.array_equal(np.asarray(o["coords"].real), want[i]): bad += 1
torch.cuda.synchronize(); free1, _ = torch.cuda.mem_get_info()
print(f"host stacks, {n} frames: {(time.perf_counter() - t0) / n * 1e3:.2f} ms per frame; first 6 frames differing from the resident run: {bad}; device free {free1 - free0:+d} B")
PY
