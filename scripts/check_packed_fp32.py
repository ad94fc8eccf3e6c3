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
import os
import shutil


def _llvm_bin() -> Path:
    """The ROCm LLVM tool directory: $ROCM_PATH, the hipcc on PATH (or $HIPCC), /opt/rocm -- in that order."""
    roots = [os.environ.get("ROCM_PATH")]
    hipcc = os.environ.get("HIPCC") or shutil.which("hipcc")
    if hipcc:
        roots.append(str(Path(hipcc).resolve().parents[1]))
    roots.append("/opt/rocm")
    for r in roots:
        if r and (Path(r) / "lib" / "llvm" / "bin" / "llvm-objdump").exists():
            return Path(r) / "lib" / "llvm" / "bin"
    raise SystemExit("[packed-fp32] cannot find ROCm's llvm-objdump / clang-offload-bundler (set ROCM_PATH): the gate cannot run, refusing to pass")


LLVM = _llvm_bin()
ARCH = os.environ.get("CT_ARCH", "gfx950")
TARGET = f"hipv4-amdgcn-amd-amdhsa--{ARCH}"


class GateError(RuntimeError):
    """device code is present but could not be inspected: the gate must fail, not pass"""
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
            raise GateError(f"{obj.name}: has a .hip_fatbin section but no {TARGET} code object could be unbundled ({r.stderr.strip()[:200]})")
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
    inspected = 0
    for o in objs:
        if o.stem in ALLOWED:
            continue
        if device_disassembly(o) is not None:
            inspected += 1
        f = packed_fp32_by_kernel(o)
        if f:
            bad[o.name] = f
    if objs and not inspected and any(o.stem not in ALLOWED for o in objs):
        raise GateError("no device code could be inspected in any object: the gate did not run")
    return bad


if __name__ == "__main__":
    try:
        bad = check(sys.argv[1:])
    except (GateError, subprocess.CalledProcessError) as e:
        print(f"[packed-fp32] gate error: {e}", file=sys.stderr)
        sys.exit(2)
    for name, f in bad.items():
        for k, c in f.items():
            print(f"[packed-fp32] {name}: {c} instruction(s) in {k}", file=sys.stderr)
    if bad:
        sys.exit(1)
    print("[packed-fp32] none outside ct_unet")
