#!/bin/bash
# round 6, call 8: small matches as one persistent launch -- bit-identity test, then timing per grid size
export GPU_MAX_HW_QUEUES=16
mkdir -p gpurun_out
python -m pytest tests/test_gpu_match.py -m gpu -x -q -k "three_launch_em or 600_cells or end_to_end or batched" 2>&1 | tail -6 > gpurun_out/r06_c8_tests.txt
{ CT_EM_PERSISTENT=0 python scripts/probe/em_small_g.py; for g in 1 2 4 8; do CT_EM_SMALL_G=$g python scripts/probe/em_small_g.py; done; } 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_c8_em_small.txt
