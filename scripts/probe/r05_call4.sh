set -u
cd $GRAFT_REPO_ROOT
bash scripts/probe/wino_ceiling.sh 2>&1 | tee gpurun_out/r05_wino_ceiling.txt
timeout 300 python scripts/microbench.py unet --layers 2>&1 | grep -v amdgpu | tail -16 | tee -a gpurun_out/r05_wino_ceiling.txt
timeout 300 python scripts/microbench.py match 2000 2>&1 | grep -v amdgpu | tail -3
timeout 900 python -m pytest tests/test_gpu_bench.py -m gpu -x -q 2>&1 | tail -5
