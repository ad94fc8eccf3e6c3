# Does a storm of wave launches by ANOTHER process -- kernels without any arithmetic -- corrupt the packed-fp32 victim?  (GPU box)
cd $GRAFT_REPO_ROOT
run() {   # $1 label, rest: wave_storm arguments
  local label=$1; shift
  rm -f /tmp/storm.log
  ( timeout 40 scripts/probe/wave_storm "$@" > /tmp/storm.log 2>&1 ) &
  for i in $(seq 1 40); do grep -q running /tmp/storm.log 2>/dev/null && break; sleep 0.25; done
  sleep 0.5
  printf "%-64s " "$label"
  scripts/probe/pk_mfma_hazard victimonly 60
  wait; tail -1 /tmp/storm.log | sed 's/^/      /'
}
printf "%-64s " "alone"; scripts/probe/pk_mfma_hazard victimonly 60
run "empty kernel, 30000 x 256 threads"                  6 0 30000 256
run "empty kernel, 30000 x 64 threads"                   6 0 30000 64
run "empty kernel, 1000 x 256 threads"                   6 0 1000 256
run "36-KB LDS kernel, 30000 x 256"                      6 1 30000 256
run "one barrier, 30000 x 256"                           6 3 30000 256
run "sleeping kernel (~20 us per wave), 30000 x 256"     6 2 30000 256
run "sleeping kernel, 1024 x 256 (one wave per SIMD)"    6 2 1024 256
