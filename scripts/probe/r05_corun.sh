# round 5, first GPU call: what the U-Net loses beside the other streams (corun.py), the frame loop's kernel stats, an SQ pass of the loop
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
timeout 300 python scripts/probe/corun.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_corun_plain.txt
timeout 300 python scripts/probe/corun.py --mask-cus 32 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_corun_mask32.txt
timeout 300 python scripts/probe/corun.py --mask-cus 32 --mask-unet 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_corun_mask32_split.txt
timeout 200 python scripts/probe/seqonly.py 2>&1 | tail -1 | tee gpurun_out/r05_seqonly.txt
bash scripts/prof.sh frameseq_r05 $GRAFT_REPO_ROOT/scripts/probe/seqonly.py | head -3
bash scripts/prof_sq.sh seqB_r05 "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" $GRAFT_REPO_ROOT/scripts/probe/seqonly.py | tail -12
cp /tmp/sq_seqB_r05.log gpurun_out/r05_seqB.log 2>/dev/null
# does the counter pass serialise the streams?  kernel durations inside the counter run vs the plain trace
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
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r05_base.json 2> gpurun_out/bench_r05_base.err; python scripts/probe/pick.py gpurun_out/bench_r05_base.json
