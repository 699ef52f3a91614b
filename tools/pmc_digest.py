#!/usr/bin/env python3
"""Digest rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into profiles/r01_traffic.json.

usage: tools/pmc_digest.py <dir of FETCH_SIZE pass> <dir of WRITE_SIZE pass> <kernel substring> [out.json]

Per launch of the named kernel: HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE (MI355X_MICROARCH.md, HBM section:
on gfx950 FETCH_SIZE tallies 128-B requests at 64 B for wide coalesced reads; WRITE_SIZE is exact for 16-B stores).
rocprofv3 reports both counters in KiB.  Unit check carried in the output: gauss_fill touches every 32-byte pixel
record of the x_t|cond buffer (8 floats per pixel, 3 written) and reads nothing, so its WRITE_SIZE must be
B*H*W*32 bytes (16*256*256*32 B = 32768 KiB at configs[1]).
"""
import csv
import glob
import json
import os
import sys


def load(d, counter):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                if r["Counter_Name"] == counter:
                    rows.append((r["Kernel_Name"], float(r["Counter_Value"])))
    return rows


def main():
    fdir, wdir, pat = sys.argv[1:4]
    out = sys.argv[4] if len(sys.argv) > 4 else None
    fetch, write = load(fdir, "FETCH_SIZE"), load(wdir, "WRITE_SIZE")
    f = [v for k, v in fetch if pat in k]
    w = [v for k, v in write if pat in k]
    assert f and w, "kernel %r not found (%d fetch rows, %d write rows)" % (pat, len(fetch), len(write))
    name = next(k for k, _ in fetch if pat in k)
    # unit calibration: gauss_fill writes exactly 4 bytes per element and reads nothing
    gw = [v for k, v in write if "gauss_fill" in k]
    res = {
        "kernel": name.split("(")[0],
        "population": "all %d launches of the kernel in the profiled command" % len(f),
        "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE --output-format csv -- python3 bench.py "
                   "--steps 2 --warmup 1 --no-cpu-baseline --no-roofline (two separate passes)",
        "fetch_size_bytes_raw": 1024.0 * sum(f) / len(f),
        "write_size_bytes": 1024.0 * sum(w) / len(w),
        "hbm_bytes_per_launch": 1024.0 * (2.0 * sum(f) / len(f) + sum(w) / len(w)),
        "unit_check": {"gauss_fill_write_size_KiB": gw[:1], "expected_KiB": 16 * 256 * 256 * 32 / 1024},
        "correction": "gfx950 FETCH_SIZE counts 64 B per 128-B request on wide coalesced reads "
                      "(MI355X_MICROARCH.md, HBM): read bytes = 2 x raw",
    }
    if out:
        json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
