set -u
cd $GRAFT_REPO_ROOT
timeout 300 python scripts/microbench.py goodprior 600 2>&1 | grep -v amdgpu | tail -1
CT_EM_PERSISTENT=0 timeout 300 python scripts/microbench.py goodprior 600 2>&1 | grep -v amdgpu | tail -1
timeout 300 python scripts/microbench.py match 600 2>&1 | grep -v amdgpu | tail -1
CT_EM_PERSISTENT=0 timeout 300 python scripts/microbench.py match 600 2>&1 | grep -v amdgpu | tail -1
timeout 200 python scripts/probe/seqonly.py 96 2>&1 | tail -1
CT_EM_PERSISTENT=0 timeout 200 python scripts/probe/seqonly.py 96 2>&1 | tail -1
timeout 900 python -m pytest tests/test_gpu_match.py -m gpu -x -q 2>&1 | tail -4
