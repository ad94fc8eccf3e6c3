#!/bin/bash
# round 6, call 3: fused pair step 2 (buffer-load gather, pk split, max3 pool) + the generic epilogue's fast paths -- parity, per-layer A/B
export GPU_MAX_HW_QUEUES=16
mkdir -p gpurun_out
python -m pytest tests/test_gpu_unet.py tests/test_gpu_unet_modes.py -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r06_c3_tests.txt
bash scripts/probe/ab_layers.sh r05 shipped > gpurun_out/r06_c3_layers.txt 2>&1
bash scripts/probe/ab_arch.sh r05 shipped > gpurun_out/r06_c3_arch.txt 2>&1
