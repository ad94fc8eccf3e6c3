#!/bin/bash
# round 6, call 22: the FFN's dense layers on the exact-fp32 MFMA -- bit-identity with the vector form, the match tests, timing alone and in the frame loop
export GPU_MAX_HW_QUEUES=16
mkdir -p gpurun_out
python -m pytest tests/test_gpu_match.py -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r06_c22_tests.txt
for rep in 1 2; do for v in 0 1; do
  echo "== CT_GEMM_VALU=$v (pass $rep)"
  CT_GEMM_VALU=$v python scripts/microbench.py goodprior 600 2>&1 | grep -v amdgpu.ids | tail -3
  CT_GEMM_VALU=$v python scripts/probe/seqonly.py 96 2>&1 | grep -v amdgpu.ids | tail -1
done; done > gpurun_out/r06_c22_gemm.txt 2>&1
