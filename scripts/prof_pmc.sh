#!/bin/bash
# usage: scripts/prof_pmc.sh <outname> <counter> <python script + args...>  -- one PMC counter per pass (own run, kernel-trace only)
set -u
name=$1; ctr=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_$name
timeout 400 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_$name -o $name -- python "$@" > /tmp/pmc_$name.log 2>&1
tail -2 /tmp/pmc_$name.log
f=$(find /tmp/pmc_$name -name "*counter_collection.csv" 2>/dev/null | head -1)
if [ -n "$f" ] && [ -f "$f" ]; then
  mkdir -p $GRAFT_REPO_ROOT/gpurun_out/prof
  python - "$f" "$GRAFT_REPO_ROOT/gpurun_out/prof/${name}_${ctr}.csv" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    k = (r.get("Kernel_Name", "?"), r.get("Counter_Name", "?"))
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += float(r.get("Counter_Value", 0) or 0)
with open(sys.argv[2], "w") as f:
    f.write("Kernel_Name,Counter_Name,Dispatches,Sum,AveragePerDispatch\n")
    for (kn, cn), (n, s) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        f.write(f"\"{kn}\",{cn},{n},{s},{s / n}\n")
print(open(sys.argv[2]).read()[:1500])
PY
else
  echo "no counter csv"; find /tmp/pmc_$name | head
fi
