"""Static instruction budget of one kernel from the device assembly, per phase (phases = the stretches between s_barrier instructions):
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only ct_unet.hip -o /tmp/ct_unet.s
    python scripts/isa_budget.py /tmp/ct_unet.s conv_l0l1_fused_kernel [--top 12]
Classes: MFMA, VALU (other v_*), of which packed / cvt / DPP / cndmask / 64-bit address math, SALU (s_* ALU), s_waitcnt, branches, exec-mask
ops (s_*_saveexec, s_or/s_and/s_andn2 on exec: the cost of divergent `if`s), LDS (ds_*), VMEM loads / stores, SMEM loads.
A static count: loops count once (the split kernels' chunk loop is marked by its backward branch), predicated-off code counts fully."""
import collections
import re
import sys

path, kname = sys.argv[1], sys.argv[2]
top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 0
lines = open(path).read().splitlines()
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(kname) + r"\w*:", l) and all(t in l for t in sys.argv[3:] if not t.startswith("--") and not t.isdigit()))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".amdhsa_kernel") or lines[i].startswith("\t.section"))
print(lines[start].split(":")[0])


def classify(op, rest):
    c = []
    if op.startswith("v_mfma") or op.startswith("v_smfmac"):
        return ["MFMA"]
    if op.startswith("v_"):
        c.append("VALU")
        if op.startswith("v_pk_"): c.append("valu:packed")
        if op.startswith("v_cvt"): c.append("valu:cvt")
        if "dpp" in rest or op.endswith("_dpp"): c.append("valu:dpp")
        if op.startswith("v_cndmask"): c.append("valu:cndmask")
        if op.startswith("v_cmp") or op.startswith("v_cmpx"): c.append("valu:cmp")
        if "u64" in op or "i64" in op or op.startswith("v_lshl_add_u64") or op.startswith("v_mad_u64") or op.startswith("v_addc") or op.startswith("v_add_co"): c.append("valu:addr64")
        if op.startswith("v_accvgpr") or op.startswith("v_mov"): c.append("valu:mov")
        if op.startswith(("v_max", "v_min")): c.append("valu:minmax")
        if op.startswith(("v_fma", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_mac", "v_fmac", "v_pk_fma", "v_pk_mul", "v_pk_add")): c.append("valu:fp32-arith")
        return c
    if op.startswith("ds_"):
        return ["LDS", "lds:read" if ("read" in op or "load" in op or "bpermute" in op or "permute" in op) else "lds:write"]
    if op.startswith(("global_load", "buffer_load", "flat_load", "scratch_load")):
        return ["VMEM_LOAD"]
    if op.startswith(("global_store", "buffer_store", "flat_store", "scratch_store", "global_atomic", "buffer_atomic", "flat_atomic")):
        return ["VMEM_STORE"]
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return ["SMEM"]
    if op == "s_waitcnt" or op.startswith("s_waitcnt"):
        return ["s_waitcnt"]
    if op == "s_barrier":
        return ["s_barrier"]
    if op.startswith(("s_cbranch", "s_branch")):
        return ["branch"]
    if op.startswith(("s_nop", "s_sleep", "s_setprio", "s_endpgm", "s_sethalt", "s_setreg", "s_getreg", "s_sendmsg", "s_code_end", "s_inst_prefetch", "s_clause", "s_delay")):
        return ["misc"]
    if op.startswith("s_"):
        c = ["SALU"]
        if "saveexec" in op or "exec" in rest: c.append("salu:exec-mask")
        if op.startswith(("s_mul", "s_add", "s_sub", "s_lshl", "s_lshr", "s_ashr", "s_addc", "s_subb", "s_mulk", "s_addk")): c.append("salu:arith")
        if op.startswith("s_cmp") or op.startswith("s_cselect") or op.startswith("s_bitcmp"): c.append("salu:cmp/select")
        if op.startswith(("s_and", "s_or", "s_xor", "s_not", "s_andn2", "s_orn2")) and "exec" not in rest and "saveexec" not in op: c.append("salu:mask-logic")
        if op.startswith(("s_mov", "s_cmov")): c.append("salu:mov")
        return c
    return ["other"]


phases = [collections.Counter()]
labels = {}
back = []
n = 0
for i in range(start + 1, end):
    l = lines[i].split(";")[0].strip()
    if not l or l.startswith("."):
        m = re.match(r"^(\.LBB\w+):", l)
        if m:
            labels[m.group(1)] = (len(phases) - 1, n)
        continue
    m = re.match(r"^(\.LBB\w+):", l)
    if m:
        labels[m.group(1)] = (len(phases) - 1, n); continue
    parts = l.split(None, 1)
    op, rest = parts[0], (parts[1] if len(parts) > 1 else "")
    n += 1
    for c in classify(op, rest):
        phases[-1][c] += 1
    if op.startswith(("s_cbranch", "s_branch")) and rest.strip() in labels:
        back.append((rest.strip(), labels[rest.strip()], len(phases) - 1, n))
    if op == "s_barrier":
        phases.append(collections.Counter())
tot = collections.Counter()
for p in phases:
    tot.update(p)
keys = ["MFMA", "VALU", "SALU", "LDS", "VMEM_LOAD", "VMEM_STORE", "SMEM", "s_waitcnt", "branch", "misc"]
print(f"{'phase':>6s} " + " ".join(f"{k:>10s}" for k in keys))
for i, p in enumerate(phases):
    print(f"{i:6d} " + " ".join(f"{p[k]:10d}" for k in keys))
print(f"{'total':>6s} " + " ".join(f"{tot[k]:10d}" for k in keys))
print("sub-classes (whole kernel):", {k: v for k, v in sorted(tot.items()) if ":" in k})
for i, p in enumerate(phases):
    sub = {k: v for k, v in sorted(p.items()) if ":" in k and v >= 8}
    print(f"  phase {i}: {sub}")
if back:
    print("backward branches (loops; the body counts once above):", [(lab, f"phase {a[0]} -> phase {ph}", f"{cnt - a[1]} instructions") for lab, a, ph, cnt in back])
if tot["MFMA"]:
    print(f"non-MFMA VALU + SALU + LDS per MFMA (static): {(tot['VALU'] + tot['SALU'] + tot['LDS']) / tot['MFMA']:.2f}")
