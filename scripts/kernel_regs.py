"""Compact per-kernel resource table (VGPRs, spills, scratch, occupancy, LDS) from hipcc's -Rpass-analysis=kernel-resource-usage.
usage: python scripts/kernel_regs.py 3deecelltracker_amd/csrc/ct_unet.hip [name filter] [extra hipcc flags...]"""
import re, subprocess, sys
src = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o", "/tmp/_regs.o",
       "-Rpass-analysis=kernel-resource-usage"] + sys.argv[3:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None; rows = []
for line in out.splitlines():
    m = re.search(r"remark: (.*?) \[-Rpass", line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": subprocess.run(["c++filt", t.split(":", 1)[1].strip()], capture_output=True, text=True).stdout.strip()}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1); cur[k.strip()] = v.strip()
for r in rows:
    n = re.sub(r"\(anonymous namespace\)::|void ", "", r["name"]); n = re.sub(r"\(.*$", "", n)
    if flt in n:
        print(f"{n:58s} VGPR {r.get('VGPRs','?'):>4} AGPR {r.get('AGPRs','?'):>3} spill {r.get('VGPR Spill','?'):>3} scratch {r.get('ScratchSize [bytes/lane]','?'):>4} occ {r.get('Occupancy [waves/SIMD]','?'):>2} LDS {r.get('LDS Size [bytes/block]','?')}")
