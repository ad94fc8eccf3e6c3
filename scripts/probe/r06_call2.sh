#!/bin/bash
# round 6, call 2: the rewritten fused first pair -- parity, then per-layer A/B against the round-5 library
export GPU_MAX_HW_QUEUES=16
mkdir -p gpurun_out
python -m pytest tests/test_gpu_unet.py tests/test_gpu_unet_modes.py -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r06_c2_tests.txt
bash scripts/probe/ab_layers.sh r05 shipped > gpurun_out/r06_c2_layers.txt 2>&1
