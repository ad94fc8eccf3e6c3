#!/bin/bash
# round 6, call 4: packed fp32 (epilogues + staging scale) against scalar fp32 in the conv kernels -- r05 | shipped (scalar, no SLP) | pk | slp (scalar source, compiler packs)
export GPU_MAX_HW_QUEUES=16
mkdir -p gpurun_out
bash scripts/probe/ab_layers.sh r05 shipped pk slp > gpurun_out/r06_c4_layers.txt 2>&1
python -m pytest tests/test_gpu_unet.py tests/test_gpu_unet_modes.py -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r06_c4_tests.txt
