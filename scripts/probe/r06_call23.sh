#!/bin/bash
# round 6, call 23: FFN front half alone, matrix-core GEMM against the vector form
export GPU_MAX_HW_QUEUES=16
mkdir -p gpurun_out
for rep in 1 2; do for v in 0 1; do echo "== CT_GEMM_VALU=$v (pass $rep)"; CT_GEMM_VALU=$v python scripts/microbench.py match 600 2>&1 | grep -v amdgpu.ids | head -6; done; done > gpurun_out/r06_c23_ffn.txt 2>&1
