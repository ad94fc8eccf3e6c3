#!/bin/bash
# usage: scripts/prof_sq.sh <outname> "<counter list>" <python script + args...>  -- one PMC pass (own run, kernel-trace only), per-kernel sums
set -u
name=$1; ctrs=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/sq_$name
timeout 400 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d /tmp/sq_$name -o $name -- python "$@" > /tmp/sq_$name.log 2>&1
tail -2 /tmp/sq_$name.log
f=$(find /tmp/sq_$name -name "*counter_collection.csv" 2>/dev/null | head -1)
if [ -n "$f" ] && [ -f "$f" ]; then
  mkdir -p $GRAFT_REPO_ROOT/gpurun_out/prof
  python - "$f" "$GRAFT_REPO_ROOT/gpurun_out/prof/${name}_sq.csv" <<'PY'
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    kn = re.sub(r"\(anonymous namespace\)::", "", r.get("Kernel_Name", "?")); kn = re.sub(r"^void ", "", kn); kn = re.sub(r"\(.*$", "", kn)
    k = (kn, r.get("Counter_Name", "?"))
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += float(r.get("Counter_Value", 0) or 0)
with open(sys.argv[2], "w") as f:
    f.write("Kernel_Name,Counter_Name,Dispatches,Sum,AveragePerDispatch\n")
    for (kn, cn), (n, s) in agg.items():
        f.write(f"\"{kn}\",{cn},{n},{s},{s / n}\n")
# pivot for the log
kernels = collections.OrderedDict()
for (kn, cn), (n, s) in agg.items():
    kernels.setdefault(kn, {})[cn] = s / n
for kn, d in kernels.items():
    if kn.startswith("conv"):
        print(kn, {k: f"{v:.3g}" for k, v in d.items()})
PY
else
  echo "no counter csv"; find /tmp/sq_$name | head
fi
