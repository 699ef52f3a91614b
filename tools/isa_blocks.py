#!/usr/bin/env python3
"""Per-basic-block instruction mix of one kernel in a hipcc -S listing:  tools/isa_blocks.py file.s <line of kernel label>"""
import re, sys, collections
lines = open(sys.argv[1]).read().split("\n")
start = int(sys.argv[2])
blocks, cur = [], None
for i in range(start, len(lines)):
    l = lines[i].strip()
    if l.startswith(".Lfunc_end"):
        break
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m or cur is None:
        cur = {"name": m.group(1) if m else "entry", "line": i + 1, "c": collections.Counter(), "br": []}
        blocks.append(cur)
        if m:
            continue
    if not l or l.startswith(";") or l.startswith("."):
        continue
    op = l.split()[0]
    if op.startswith("v_mfma"): k = "mfma"
    elif op.startswith("v_"):
        k = "valu"
        cur["c"]["v:" + op] += 1
    elif op.startswith("ds_"): k = "lds"
    elif op.startswith("buffer_") or op.startswith("global_"): k = "vmem"
    elif op.startswith("s_cbranch") or op.startswith("s_branch"):
        k = "salu"; cur["br"].append(l.split()[-1])
    elif op.startswith("s_"): k = "salu"
    else: k = "other"
    cur["c"][k] += 1
for b in blocks:
    c = b["c"]
    if sum(c[k] for k in ("mfma", "valu", "lds", "vmem")) < 8: continue
    top = sorted(((v, k[2:]) for k, v in c.items() if k.startswith("v:")), reverse=True)[:8]
    print(f"{b['name']:12s} @{b['line']:7d} mfma {c['mfma']:4d} valu {c['valu']:5d} lds {c['lds']:4d} vmem {c['vmem']:4d} salu {c['salu']:4d} -> {','.join(b['br'])}  | " + " ".join(f"{k}:{v}" for v, k in top))
