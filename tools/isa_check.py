#!/usr/bin/env python3
"""Build-time ISA check for the gfx950 buffer-store data hazard (profiles/r01_store_hazard.txt).

A `buffer_store_dwordx3/x4 vdata, voff, rsrc, sN offen` (soffset in an SGPR) followed directly by a VALU write to
one of its data registers stores the NEW value in some lanes: hipcc's hazard recogniser (ROCm 7.2) exempts the
SGPR-soffset form.  csrc/common.h `buf_store4` therefore issues `s_nop 1` tied to the data registers; nothing in
the language forces the compiler to keep that nop adjacent, so this script re-checks the generated code:

    every wide buffer store whose soffset is a register must be followed -- before any instruction that writes
    one of its data VGPRs -- by an s_nop (or any other instruction sequence that is not a VALU/VMEM write to them).

usage: tools/isa_check.py [file.hip ...]     (default: every csrc/*.hip); exit status 1 on a violation.
Used by tests/test_abi.py (CPU, no GPU needed: hipcc cross-compiles).
"""
from __future__ import annotations

import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "conditional-diffusion-model-for-compression_amd", "csrc")
STORE = re.compile(r"^\s*buffer_store_dwordx([34])\s+v\[(\d+):(\d+)\],\s*\S+,\s*s\[\d+:\d+\],\s*(s\d+|m0|vcc_lo|vcc_hi)\s+offen")
VREG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def device_asm(src: str, extra=()) -> str:
    r = subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S", src, "-o", "-",
                        *extra], capture_output=True, text=True, cwd=os.path.dirname(src))
    if r.returncode:
        raise RuntimeError(r.stderr[-3000:])
    return r.stdout


def written_vgprs(line: str) -> set[int]:
    """VGPRs the instruction's FIRST operand names (the destination of VALU / load instructions)."""
    parts = line.strip().split(None, 1)
    if len(parts) < 2 or parts[0].startswith(("s_", "buffer_store", "global_store", "ds_write", "ds_store", ";")):
        return set()
    first = parts[1].split(",")[0]
    m = VREG.search(first)
    if not m:
        return set()
    if m.group(1) is not None:
        return {int(m.group(1))}
    return set(range(int(m.group(2)), int(m.group(3)) + 1))


def check(asm: str) -> tuple[int, list[str]]:
    lines = [ln for ln in asm.splitlines() if ln.strip() and not ln.strip().startswith((";", ".", "//")) and not ln.rstrip().endswith(":")]
    bad, stores = [], 0
    for i, ln in enumerate(lines):
        m = STORE.match(ln)
        if not m:
            continue
        stores += 1
        data = set(range(int(m.group(2)), int(m.group(3)) + 1))
        nxt = lines[i + 1] if i + 1 < len(lines) else ""
        if nxt.strip().startswith("s_nop"):
            continue
        if written_vgprs(nxt) & data:
            bad.append(f"{ln.strip()}   ->   {nxt.strip()}")
    return stores, bad


def main(argv):
    files = argv or sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    rc = 0
    for f in files:
        stores, bad = check(device_asm(f))
        print(f"{os.path.basename(f)}: {stores} wide buffer stores with a register soffset, {len(bad)} unprotected")
        for b in bad:
            print("   HAZARD:", b)
            rc = 1
    return rc


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
