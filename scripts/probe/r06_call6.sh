#!/bin/bash
# round 6, call 6: the whole GPU suite on the current tree + the bench line
export GPU_MAX_HW_QUEUES=16
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r06_c6_tests.txt
python bench.py --steps 20 --warmup 3 > gpurun_out/r06_c6_bench.json 2> gpurun_out/r06_c6_bench.err
