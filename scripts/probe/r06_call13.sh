#!/bin/bash
# round 6, call 13: stream priorities again, now that the watershed's sweeps are 1024 workgroups of 512 threads (round 5: 0,-1,-1 was the best of nine)
export GPU_MAX_HW_QUEUES=16
mkdir -p gpurun_out
for rep in 1 2; do for p in "0,-1,-1" "0,0,-1" "0,-1,0" "0,0,0"; do
  echo "== CT_SEQ_PRIO=$p (pass $rep)"
  CT_SEQ_PRIO=$p python scripts/probe/seqonly.py 96 2>&1 | grep -v amdgpu.ids | tail -1
done; done > gpurun_out/r06_c13_prio.txt 2>&1
