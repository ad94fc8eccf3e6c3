#!/bin/bash
# round 6, call 12: grid-stride watershed sweeps without the per-thread division, defaults block 1024 / grid 1024 -- A/B in the frame loop, tests, bench
export GPU_MAX_HW_QUEUES=16
mkdir -p gpurun_out
for rep in 1 2; do for cfg in "256 0" "1024 0" "1024 1024" "1024 512" "512 1024"; do set -- $cfg
  echo "== CT_WS_BLOCK=$1 CT_WS_GRID=$2 (pass $rep)"
  CT_WS_BLOCK=$1 CT_WS_GRID=$2 python scripts/microbench.py watershed 2>&1 | grep -v amdgpu.ids | head -1
  CT_WS_BLOCK=$1 CT_WS_GRID=$2 python scripts/probe/seqonly.py 96 2>&1 | grep -v amdgpu.ids | tail -1
done; done > gpurun_out/r06_c12_ws_grid.txt 2>&1
python -m pytest tests/test_watershed.py tests/test_watershed_pin.py tests/test_segment.py tests/test_gpu_bench.py tests/test_legacy_tracker.py -m gpu -x -q 2>&1 | tail -3 > gpurun_out/r06_c12_tests.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r06_c12_bench.json 2> gpurun_out/r06_c12_bench.err
