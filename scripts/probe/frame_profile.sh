#!/bin/bash
# Where a chained frame's time goes: rocprofv3 kernel stats of `microbench.py frame` (the dependent per-frame chain), its own stage timing, and a
# sweep of the pipelined benchmark's match schedule.  Run on the GPU box: bash scripts/probe/frame_profile.sh <tag>
tag=${1:-r04}
cd $GRAFT_REPO_ROOT
bash scripts/prof.sh frame_$tag $GRAFT_REPO_ROOT/scripts/microbench.py frame > /dev/null 2>&1
python - <<PY
import csv, re
rows = list(csv.DictReader(open("gpurun_out/prof/frame_${tag}_kernel_stats.csv")))
print("total kernel ms in the trace", sum(int(r["TotalDurationNs"]) for r in rows) / 1e6)
for r in rows[:48]:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Name"]); n = re.sub(r"\(.*", "", n)[:60]
    print(f"{n:60s} {int(r['Calls']):5d} {float(r['AverageNs'])/1e3:9.1f} us  {float(r['Percentage']):5.2f}%")
PY
timeout 120 python scripts/microbench.py frame 2>&1 | grep -v amdgpu | tail -8
for opt in "" "--match-workers 2 --match-batch 16" "--match-batch 64" "--match-workers 2 --match-batch 32"; do
  timeout 200 python bench.py --no-cpu-baseline --no-realistic-pass $opt 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$opt ->', d['value'], 'volumes/s', d['ms_per_step'], 'ms/step')"
done
