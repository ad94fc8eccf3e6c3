#!/bin/bash
# usage: scripts/prof.sh <outname> <python script + args...>   (runs on the GPU box, bounded)
set -u
name=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$name
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -o $name -- python "$@" > /tmp/prof_$name.log 2>&1
tail -2 /tmp/prof_$name.log
f=$(find /tmp/prof_$name -name "*kernel_stats.csv" 2>/dev/null | head -1)
if [ -n "$f" ] && [ -f "$f" ]; then
  mkdir -p $GRAFT_REPO_ROOT/gpurun_out/prof
  cp "$f" $GRAFT_REPO_ROOT/gpurun_out/prof/${name}_kernel_stats.csv
  head -25 "$f"
else
  echo "no kernel_stats.csv produced"; find /tmp/prof_$name | head
fi
