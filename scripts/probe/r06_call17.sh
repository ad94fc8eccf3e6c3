#!/bin/bash
# round 6, call 17: the co-run attribution of round 5 (scripts/probe/corun.py) again, with this round's conv kernels and the watershed's sweeps as 1024 x 512 workgroups;
# and with the round-5 issue shape of the sweeps (CT_WS_BLOCK=256 CT_WS_GRID=0) in the same call
export GPU_MAX_HW_QUEUES=16
mkdir -p gpurun_out
{ echo "== shipped (sweeps 1024 workgroups x 512 threads)"; python scripts/probe/corun.py --frames 48 2>&1 | grep -v amdgpu.ids
  echo "== CT_WS_BLOCK=256 CT_WS_GRID=0 (round 5's 32768 x 256)"; CT_WS_BLOCK=256 CT_WS_GRID=0 python scripts/probe/corun.py --frames 48 2>&1 | grep -v amdgpu.ids; } > gpurun_out/r06_c17_corun.txt
