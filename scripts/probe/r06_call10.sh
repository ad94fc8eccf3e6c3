#!/bin/bash
# round 6, call 10: watershed sweeps at 1024 threads per workgroup as the default -- segment / watershed / frame tests, the fixed dynamic-range test, bench
export GPU_MAX_HW_QUEUES=16
mkdir -p gpurun_out
python -m pytest tests/test_gpu_unet.py -m gpu -x -q -k "bound_of_the_first" 2>&1 | tail -12 > gpurun_out/r06_c10_tests.txt
python -m pytest tests/test_watershed.py tests/test_watershed_pin.py tests/test_segment.py tests/test_gpu_bench.py tests/test_legacy_tracker.py -m gpu -x -q 2>&1 | tail -4 >> gpurun_out/r06_c10_tests.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r06_c10_bench.json 2> gpurun_out/r06_c10_bench.err
