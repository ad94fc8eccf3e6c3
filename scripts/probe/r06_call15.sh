#!/bin/bash
# round 6, call 15: the fused first pair as a PERSISTENT grid (CT_FUSE_PERSIST workgroups per CU; 0 = one workgroup per tile) -- alone, and inside the frame loop
export GPU_MAX_HW_QUEUES=16
mkdir -p gpurun_out
python -m pytest tests/test_gpu_unet.py -m gpu -x -q 2>&1 | tail -3 > gpurun_out/r06_c15_tests.txt
for rep in 1 2; do for pc in 0 4 3 2; do
  echo "== CT_FUSE_PERSIST=$pc (pass $rep)"
  CT_FUSE_PERSIST=$pc python scripts/microbench.py unet --layers 2>&1 | grep -v amdgpu.ids | tr '\n' ' ' | sed 's/ms\/vol (1 launches)//g; s/  */ /g' | cut -c1-330; echo
  CT_FUSE_PERSIST=$pc python scripts/probe/seqonly.py 96 2>&1 | grep -v amdgpu.ids | tail -1
done; done > gpurun_out/r06_c15_persist.txt 2>&1
