#!/bin/bash
# round 6, call 1: the housekeeping commit on the GPU (match / parallel / frame tests) + today's per-layer baseline
export GPU_MAX_HW_QUEUES=16
mkdir -p gpurun_out
python -m pytest tests/test_gpu_match.py tests/test_gpu_multirank.py tests/test_gpu_bench.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r06_c1_tests.txt
bash scripts/probe/ab_layers.sh shipped > gpurun_out/r06_c1_layers.txt 2>&1
python bench.py --steps 20 --warmup 3 > gpurun_out/r06_c1_bench.json 2> gpurun_out/r06_c1_bench.err
