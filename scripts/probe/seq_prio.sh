# frame loop under different stream priorities (U-Net, watershed, match): python scripts/probe/seqonly.py with CT_SEQ_PRIO
cd $GRAFT_REPO_ROOT
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range())"
for p in "0,-1,-1" "0,0,-1" "0,1,-1" "-1,0,0" "-1,1,0" "-1,1,-1" "0,0,0" "0,1,0" "0,1,1"; do
  echo -n "CT_SEQ_PRIO=$p: "; CT_SEQ_PRIO=$p timeout 200 python scripts/probe/seqonly.py 96 2>&1 | tail -1
done
