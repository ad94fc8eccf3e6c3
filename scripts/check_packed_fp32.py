"""Build gate: no packed-fp32 VALU arithmetic (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) in any device object except ct_unet.

Why: on MI355X such instructions with LDS-fed operands returned wrong values in lanes 48-63 whenever waves of the
split conv kernels shared the SIMD.  ct_unet's own packed instructions are validated against the oracle beside their own MFMA waves;
every other translation unit must not contain any, whatever the compiler version or a future edit does.

usage: python scripts/check_packed_fp32.py [object files...]   (default: every csrc/*.o)   exit code 1 on a finding."""
from __future__ import annotations

import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
CSRC = ROOT / "3deecelltracker_amd" / "csrc"
LLVM = Path("/opt/rocm/lib/llvm/bin")
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"
ALLOWED = {"ct_unet"}
PACKED = re.compile(r"\bv_pk_(fma|mul|add)_f32\b")
KERNEL = re.compile(r"^[0-9a-f]+ <(.+)>:$")


def device_disassembly(obj: Path) -> str | None:
    """Disassembly of the gfx950 code object bundled in a host object (None if the object has no device code)."""
    with tempfile.TemporaryDirectory() as tmp:
        fat, co = Path(tmp) / "fat.bin", Path(tmp) / "dev.co"
        subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", str(obj), str(fat)], check=True)
        if not fat.exists() or fat.stat().st_size == 0:
            return None
        r = subprocess.run([str(LLVM / "clang-offload-bundler"), "--type=o", f"--targets={TARGET}", f"--input={fat}",
                            f"--output={co}", "--unbundle"], capture_output=True, text=True)
        if r.returncode != 0 or not co.exists() or co.stat().st_size == 0:
            return None
        return subprocess.run([str(LLVM / "llvm-objdump"), "-d", str(co)], check=True, capture_output=True, text=True).stdout


def packed_fp32_by_kernel(obj: Path) -> dict[str, int]:
    text = device_disassembly(obj)
    found: dict[str, int] = {}
    if text is None:
        return found
    cur = "?"
    for line in text.splitlines():
        m = KERNEL.match(line)
        if m:
            cur = m.group(1)
        elif PACKED.search(line):
            found[cur] = found.get(cur, 0) + 1
    return found


def check(objs=None) -> dict[str, dict[str, int]]:
    objs = [Path(o) for o in objs] if objs else sorted(CSRC.glob("*.o"))
    bad = {}
    for o in objs:
        if o.stem in ALLOWED:
            continue
        f = packed_fp32_by_kernel(o)
        if f:
            bad[o.name] = f
    return bad


if __name__ == "__main__":
    bad = check(sys.argv[1:])
    for name, f in bad.items():
        for k, c in f.items():
            print(f"[packed-fp32] {name}: {c} instruction(s) in {k}", file=sys.stderr)
    if bad:
        sys.exit(1)
    print("[packed-fp32] none outside ct_unet")
