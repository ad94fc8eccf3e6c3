#!/bin/bash
# round 6, call 9: the fused pair's bound-based scale over the dynamic range (new test); workgroup size of the watershed's per-voxel sweeps (256 / 512 / 1024):
# the watershed alone, bit-identity against the pins, and what the U-Net loses beside it in the frame loop
export GPU_MAX_HW_QUEUES=16
mkdir -p gpurun_out
python -m pytest tests/test_gpu_unet.py -m gpu -x -q -k "bound_of_the_first" 2>&1 | tail -4 > gpurun_out/r06_c9_tests.txt
for b in 256 512 1024 256 512 1024; do
  echo "== CT_WS_BLOCK=$b"
  CT_WS_BLOCK=$b python scripts/microbench.py watershed 2>&1 | grep -v amdgpu.ids | head -2
  CT_WS_BLOCK=$b python scripts/probe/seqonly.py 96 2>&1 | grep -v amdgpu.ids | tail -2
done > gpurun_out/r06_c9_ws_block.txt 2>&1
CT_WS_BLOCK=1024 python -m pytest tests/test_watershed.py tests/test_watershed_pin.py -m gpu -x -q 2>&1 | tail -3 >> gpurun_out/r06_c9_tests.txt
