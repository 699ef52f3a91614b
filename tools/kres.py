#!/usr/bin/env python3
"""Compile one .hip file for gfx950 and print a one-line-per-kernel resource table
(VGPR / AGPR / SGPR / spills / LDS / waves per SIMD).  Usage: tools/kres.py file.hip [extra hipcc flags]"""
import re, subprocess, sys
src, extra = sys.argv[1], sys.argv[2:]
r = subprocess.run(["hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-c", src, "-o", "/dev/null",
                    "-Rpass-analysis=kernel-resource-usage"] + extra, capture_output=True, text=True)
if r.returncode:
    print(r.stderr[-4000:]); sys.exit(1)
rows, cur = [], {}
for line in r.stderr.splitlines():
    m = re.search(r"remark:\s+(.*?):\s*(\S+)\s*\[-Rpass", line)
    if not m: continue
    k, v = m.group(1).strip(), m.group(2)
    if k == "Function Name":
        cur = {"name": subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()[:90]}
        rows.append(cur)
    else:
        cur[k] = v
print(f"{'kernel':92s} vgpr agpr sgpr vspill  lds    occ")
for c in rows:
    print(f"{c['name']:92s} {c.get('VGPRs','?'):>4} {c.get('AGPRs','?'):>4} {c.get('TotalSGPRs','?'):>4} {c.get('VGPRs Spill','?'):>6} {c.get('LDS Size [bytes/block]','?'):>6} {c.get('Occupancy [waves/SIMD]','?'):>4}")
