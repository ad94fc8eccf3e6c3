#!/bin/bash
# round 6, call 5: bias as the accumulators' initial value + out_mul folded into the BatchNorm scale (all split kernels), L1's input scale from a bound (fused pair)
export GPU_MAX_HW_QUEUES=16
mkdir -p gpurun_out
python -m pytest tests/test_gpu_unet.py tests/test_gpu_unet_modes.py -m gpu -x -q 2>&1 | tail -12 > gpurun_out/r06_c5_tests.txt
bash scripts/probe/ab_layers.sh r05 shipped > gpurun_out/r06_c5_layers.txt 2>&1
python tests/report_accuracy.py unet3_a unet3_c unet3_b > gpurun_out/r06_c5_accuracy.txt 2>&1
