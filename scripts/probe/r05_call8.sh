set -u
cd $GRAFT_REPO_ROOT
bash scripts/prof.sh frameseq2_r05 $GRAFT_REPO_ROOT/scripts/probe/seqonly.py | head -2
grep "frame sequence" /tmp/prof_frameseq2_r05.log
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
