#!/bin/bash
# round 6, call 14: the big clears (label volume x 2 per watershed, overlap counts x 3-4 per correction) -- runtime memset against the library's fill kernel, and the
# fill kernel's grid (CT_FILL_BLOCKS) beside the U-Net
export GPU_MAX_HW_QUEUES=16
mkdir -p gpurun_out
for rep in 1 2; do for cfg in "1 4096" "0 4096" "0 1024" "0 512" "0 256"; do set -- $cfg
  echo "== CT_WS_MEMSET=$1 CT_FILL_BLOCKS=$2 (pass $rep)"
  CT_WS_MEMSET=$1 CT_FILL_BLOCKS=$2 python scripts/probe/seqonly.py 96 2>&1 | grep -v amdgpu.ids | tail -1
done; done > gpurun_out/r06_c14_fill.txt 2>&1
python -m pytest tests/test_watershed_pin.py tests/test_correction.py -m gpu -x -q 2>&1 | tail -2 > gpurun_out/r06_c14_tests.txt
