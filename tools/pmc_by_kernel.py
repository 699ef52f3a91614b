#!/usr/bin/env python3
"""Mean counter value per kernel name from a rocprofv3 --pmc output directory (the *counter_collection.csv files).
usage: tools/pmc_by_kernel.py <dir> [name substring]"""
import csv, glob, os, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    with open(f, newline="") as fh:
        for r in csv.DictReader(fh):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for k, cs in sorted(acc.items()):
    if pat in k:
        print(k.split("(")[0][:110], {c: (round(sum(v) / len(v), 1), len(v)) for c, v in cs.items()})
