#!/bin/bash
# Inter-kernel gaps on the U-Net stream of the pipelined benchmark (kernel trace of a short run), printed as a small table.
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/gp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/gp -o gp -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-realistic-pass --steps 16 --warmup 2 "$@" > /tmp/gp.log 2>&1
f=$(find /tmp/gp -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
print(len(rows), "kernels; columns:", list(rows[0].keys())[:14])
by_q = collections.defaultdict(list)
for r in rows:
    by_q[r.get("Queue_Id", "?")].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
for q, ks in by_q.items():
    ks.sort()
    conv = [k for k in ks if "conv" in k[2] or "box1d" in k[2] or "tile_" in k[2] or "radix" in k[2]]
    if len(conv) < 50: continue
    busy = sum(e - s for s, e, _ in ks); span = ks[-1][1] - ks[0][0]
    gaps = [ks[i + 1][0] - ks[i][1] for i in range(len(ks) - 1)]
    gaps_s = sorted(g for g in gaps if g > 0)
    print(f"queue {q}: {len(ks)} kernels, span {span/1e6:.1f} ms, busy {busy/1e6:.1f} ms ({100*busy/span:.0f} %), gaps: n {len(gaps_s)} sum {sum(gaps_s)/1e6:.1f} ms "
          f"median {gaps_s[len(gaps_s)//2]/1e3:.1f} us p90 {gaps_s[int(len(gaps_s)*0.9)]/1e3:.1f} us max {gaps_s[-1]/1e3:.0f} us")
    big = collections.Counter()
    for i, g in enumerate(gaps):
        if g > 20000: big[(ks[i][2][:40], ks[i + 1][2][:40])] += g
    for (a, b), g in big.most_common(6): print(f"     {g/1e6:6.2f} ms of gaps between {a} -> {b}")
PY
