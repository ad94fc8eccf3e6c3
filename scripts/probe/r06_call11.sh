#!/bin/bash
# round 6, call 11: grid-stride sweeps of the watershed -- how few workgroups should a sweep use beside the U-Net?  (block 1024; grid 8192 = one slab each)
export GPU_MAX_HW_QUEUES=16
mkdir -p gpurun_out
for rep in 1 2; do for g in 0 2048 1024 512 256; do
  echo "== CT_WS_GRID=$g (pass $rep)"
  CT_WS_GRID=$g python scripts/microbench.py watershed 2>&1 | grep -v amdgpu.ids | head -1
  CT_WS_GRID=$g python scripts/probe/seqonly.py 96 2>&1 | grep -v amdgpu.ids | tail -1
done; done > gpurun_out/r06_c11_ws_grid.txt 2>&1
CT_WS_GRID=512 python -m pytest tests/test_watershed.py tests/test_watershed_pin.py tests/test_segment.py -m gpu -x -q 2>&1 | tail -3 > gpurun_out/r06_c11_tests.txt
