# round 5, call 2: tests of what the advisor fixes touched + the frame loop's kernel stats + an SQ pass of the loop (does the counter pass serialise?)
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
timeout 900 python -m pytest tests/test_watershed.py tests/test_watershed_pin.py tests/test_correction.py tests/test_gpu_hazard.py tests/test_gpu_bench.py tests/test_gpu_match.py -m gpu -x -q 2>&1 | tail -4
bash scripts/prof.sh frameseq_r05 $GRAFT_REPO_ROOT/scripts/probe/seqonly.py | head -2
bash scripts/prof_sq.sh seqB_r05 "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" $GRAFT_REPO_ROOT/scripts/probe/seqonly.py | tail -12
grep "frame sequence" /tmp/prof_frameseq_r05.log /tmp/sq_seqB_r05.log
f=$(find /tmp/sq_seqB_r05 -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections, re
if len(sys.argv) > 1 and sys.argv[1]:
    d = collections.defaultdict(list)
    rows = list(csv.DictReader(open(sys.argv[1])))
    for r in rows:
        n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); n = re.sub(r"^void ", "", n); n = re.sub(r"\(.*$", "", n)
        d[n].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    ivs = sorted(iv for v in d.values() for iv in v)
    overlap = sum(1 for a, b in zip(ivs, ivs[1:]) if b[0] < a[1])
    print("counter pass: dispatches", len(ivs), "of which start before the previous one ended:", overlap)
    for n, v in sorted(d.items(), key=lambda kv: -sum(b - a for a, b in kv[1]))[:12]:
        print(f"  {n[:70]:70s} {len(v):5d} {sum(b - a for a, b in v) / len(v) / 1e3:9.1f} us")
PY
