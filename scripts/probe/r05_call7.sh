set -u
cd $GRAFT_REPO_ROOT
bash scripts/probe/seq_prio.sh 2>&1 | grep -v amdgpu | tee gpurun_out/r05_seq_prio.txt
