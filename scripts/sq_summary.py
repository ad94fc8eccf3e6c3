"""Per-kernel digest of the SQ counter passes (scripts/prof_sq.sh) and the kernel-trace stats of the same command:
    python scripts/sq_summary.py <kernel_stats.csv> <out.json> <sq pass csv>...
For every conv kernel: launches, average duration, MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES),
effective shader clock = SQ_BUSY_CU_CYCLES / 256 CUs / duration (counter passes run a few per cent slower than plain runs), and
the wave-time shares parked (s_waitcnt / barrier) : issue-stalled : issuing (SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY over
SQ_WAVE_CYCLES).  Counter values are per-dispatch averages over all layers that use the same kernel instantiation."""
import csv, json, re, sys

stats, out, passes = sys.argv[1], sys.argv[2], sys.argv[3:]
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n); return re.sub(r"\(.*$", "", n)
dur = {}
for r in csv.DictReader(open(stats)):
    dur[short(r["Name"])] = (int(r["Calls"]), float(r["AverageNs"]))
ctr = {}
for p in passes:
    for r in csv.DictReader(open(p)):
        ctr.setdefault(r["Kernel_Name"], {})[r["Counter_Name"]] = float(r["AveragePerDispatch"])
res = {}
for k, c in ctr.items():
    if not k.startswith("conv"):
        continue
    d = {"launches_in_trace": dur.get(k, (0, 0))[0], "avg_duration_us": round(dur.get(k, (0, 0))[1] / 1e3, 1)}
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and c.get("SQ_BUSY_CU_CYCLES"):
        d["mfma_pipe_busy"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * c["SQ_BUSY_CU_CYCLES"]), 3)
    if c.get("SQ_BUSY_CU_CYCLES") and d["avg_duration_us"]:
        d["effective_clock_GHz"] = round(c["SQ_BUSY_CU_CYCLES"] / 256.0 / (d["avg_duration_us"] * 1e3), 2)
    if c.get("SQ_WAVE_CYCLES"):
        w = c["SQ_WAVE_CYCLES"]
        d["wave_time"] = {"parked_waitcnt_barrier": round(c.get("SQ_WAIT_ANY", 0) / w, 3), "issue_stalled": round(c.get("SQ_WAIT_INST_ANY", 0) / w, 3),
                          "issuing": round(c.get("SQ_ACTIVE_INST_ANY", 0) / w, 3)}
        if "SQ_WAIT_INST_LDS" in c:
            d["wave_time"]["of_which_lds_issue_stall"] = round(c["SQ_WAIT_INST_LDS"] / w, 3)
    for name in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VALU_MFMA_MOPS_F16", "SQ_INSTS_MFMA"):
        if name in c:
            d[name.lower() + "_per_dispatch"] = c[name]
    # instructions that are not MFMAs per MFMA.  SQ_INSTS_VALU counts the MFMAs too; one v_mfma_f32_16x16x32_f16 keeps the pipe busy for 16 cycles
    # (profiles/r04_mfma_ceiling.txt: 2411 TFLOP/s at 2.39 GHz on 1024 SIMDs), so the MFMA count of a dispatch is its busy cycles / 16.
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "SQ_INSTS_VALU" in c and c["SQ_VALU_MFMA_BUSY_CYCLES"] > 0:
        n_mfma = c["SQ_VALU_MFMA_BUSY_CYCLES"] / 16.0
        d["mfma_insts_per_dispatch_est"] = round(n_mfma)
        d["non_mfma_insts_per_mfma"] = round((c["SQ_INSTS_VALU"] - n_mfma + c.get("SQ_INSTS_SALU", 0.0) + c.get("SQ_INSTS_LDS", 0.0)) / n_mfma, 2)
    res[k] = d
json.dump({"_comment": __doc__, "kernels": res}, open(out, "w"), indent=1)
for k, d in res.items():
    print(k, {a: b for a, b in d.items() if not a.endswith("_per_dispatch")})
