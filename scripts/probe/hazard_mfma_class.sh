# Which MFMA instruction class, issued by ANOTHER process's waves (register operands only, nothing else in the loop), corrupts the
# packed-fp32 victim?  (GPU box; scripts/probe/pk_mfma_hazard built from pk_mfma_hazard.hip)
cd $GRAFT_REPO_ROOT
run() {   # $1 class id
  rm -f /tmp/aggr.log
  ( timeout 60 scripts/probe/pk_mfma_hazard aggr $1 7 > /tmp/aggr.log 2>&1 ) &
  for i in $(seq 1 40); do grep -q running /tmp/aggr.log 2>/dev/null && break; sleep 0.25; done
  sleep 0.5
  scripts/probe/pk_mfma_hazard victimonly 60 | sed "s/^/class $1: /"
  scripts/probe/pk_mfma_hazard victim1only 20 | sed "s/^/class $1: /"
  wait; tail -1 /tmp/aggr.log | sed 's/^/      /'
}
printf "alone:   "; scripts/probe/pk_mfma_hazard victimonly 60
for c in 0 4 3 2 5 1 6 7 8; do run $c; done
