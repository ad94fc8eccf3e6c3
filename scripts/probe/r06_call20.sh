#!/bin/bash
# round 6, call 20: the driver's form of the bench once more on whatever box this call gets (box-to-box spread of the headline)
export GPU_MAX_HW_QUEUES=16
mkdir -p gpurun_out
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r06_c20_bench.json 2> gpurun_out/r06_c20_bench.err
