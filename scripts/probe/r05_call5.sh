set -u
cd $GRAFT_REPO_ROOT
bash scripts/probe/wino_ceiling.sh 2>&1 | grep -v "coredump" | tee gpurun_out/r05_wino_ceiling.txt
timeout 300 python scripts/probe/m2000.py 2>&1 | grep -v amdgpu | tail -8
bash scripts/prof.sh m2000_r05 $GRAFT_REPO_ROOT/scripts/probe/m2000.py | head -22
timeout 900 python -m pytest tests/test_gpu_bench.py -m gpu -x -q 2>&1 | tail -5
