#!/bin/bash
# round 6, call 19: NT = 4 kernels with ONE epilogue path (shipped) against the dual path (dual4) and the round-5 library: the three architectures, patch entry point
export GPU_MAX_HW_QUEUES=16
mkdir -p gpurun_out
bash scripts/probe/ab_arch.sh r05 dual4 shipped > gpurun_out/r06_c19_arch.txt 2>&1
python -m pytest tests/test_gpu_unet.py tests/test_gpu_unet_modes.py -m gpu -x -q 2>&1 | tail -2 > gpurun_out/r06_c19_tests.txt
