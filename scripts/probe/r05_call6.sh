set -u
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_match.py -m gpu -x -q 2>&1 | tail -4
timeout 200 python scripts/probe/seqonly.py 2>&1 | tail -1
timeout 1700 python -m pytest tests/test_watershed.py tests/test_watershed_pin.py -m gpu -x -q --durations=8 2>&1 | tail -16
timeout 300 python scripts/microbench.py watershed 2>&1 | grep -v amdgpu | tail -3
