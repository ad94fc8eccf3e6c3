#!/bin/bash
# round 6, call 7: lo part of the fp16 split by v_fma_mixlo/hi_f16 (all staging), the head's sigmoid work over both channel halves (L13) -- parity, A/B against
# the previous step (step3) and round 5; cfg1 / cfg2 host profile
export GPU_MAX_HW_QUEUES=16
mkdir -p gpurun_out
python -m pytest tests/test_gpu_unet.py tests/test_gpu_unet_modes.py -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r06_c7_tests.txt
bash scripts/probe/ab_layers.sh r05 step3 shipped > gpurun_out/r06_c7_layers.txt 2>&1
bash scripts/probe/ab_arch.sh r05 step3 shipped > gpurun_out/r06_c7_arch.txt 2>&1
python tests/report_accuracy.py unet3_a > gpurun_out/r06_c7_accuracy.txt 2>&1
python scripts/probe/cfg1_host_profile.py 64 > gpurun_out/r06_c7_cfg1_profile.txt 2>&1
