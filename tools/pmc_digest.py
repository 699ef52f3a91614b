#!/usr/bin/env python3
"""Digest rocprofv3 --pmc passes into the tracked JSON files under profiles/.

  tools/pmc_digest.py traffic <dir of FETCH_SIZE pass> <dir of WRITE_SIZE pass> <kernel substring> [out.json [config]]
  tools/pmc_digest.py mfma    <dir of SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE pass> <kernel substring> <mfma-cycles-per-launch> [out.json]

traffic -- per launch of the named kernel: HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE (MI355X_MICROARCH.md, HBM section: on
gfx950 FETCH_SIZE tallies 128-B requests at 64 B for wide coalesced reads; WRITE_SIZE is exact for 16-B stores).  rocprofv3
reports both in KiB.  Unit check carried in the output: gauss_fill touches every 32-byte pixel record of the x_t|cond buffer and
reads nothing, so its WRITE_SIZE must be B*H*W*32 bytes (16*256*256*32 B = 32768 KiB at configs[1]).

mfma -- per launch: SQ_VALU_MFMA_BUSY_CYCLES (summed over the chip's SIMDs) against the cycles the chip offered,
GRBM_GUI_ACTIVE (rocprofv3 reports the SUM over the 8 XCDs) / 8 x 256 CUs x 4 SIMDs.  Unit check: the expected counter value
is the kernel's MFMA instruction count x the instruction's issue cycles (64 for v_mfma_f32_32x32x2_f32, 32 for
v_mfma_f32_32x32x16_f16); the caller passes that product per launch (averaged over the profiled launches) and the digest
reports counter / expected.

Both outputs carry csrc_sha16 (hash of the kernel sources): bench.py accepts a traffic figure only for the build it runs.
"""
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def load(d, counter):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                if r["Counter_Name"] == counter:
                    rows.append((r["Kernel_Name"], float(r["Counter_Value"])))
    return rows


def sha():
    from bench import csrc_sha16
    return csrc_sha16()


def traffic(argv):
    fdir, wdir, pat = argv[:3]
    out = argv[3] if len(argv) > 3 else None
    config = argv[4] if len(argv) > 4 else "cfg2"          # the bench.py --config the passes ran (bench.py checks it)
    fetch, write = load(fdir, "FETCH_SIZE"), load(wdir, "WRITE_SIZE")
    f = [v for k, v in fetch if pat in k]
    w = [v for k, v in write if pat in k]
    assert f and w, "kernel %r not found (%d fetch rows, %d write rows)" % (pat, len(fetch), len(write))
    name = next(k for k, _ in fetch if pat in k)
    gw = [v for k, v in write if "gauss_fill" in k]
    res = {
        "kernel": name.split("(")[0],
        "pattern": pat,
        "config": config,
        "csrc_sha16": sha(),
        "population": "all %d launches of kernels whose name contains the pattern, in the profiled command" % len(f),
        "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE --output-format csv -- python3 bench.py "
                   + ("" if config == "cfg2" else "--config %s " % config) +
                   "--steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-sample-call (two separate passes)",
        "fetch_size_bytes_raw": 1024.0 * sum(f) / len(f),
        "write_size_bytes": 1024.0 * sum(w) / len(w),
        "hbm_bytes_per_launch": 1024.0 * (2.0 * sum(f) / len(f) + sum(w) / len(w)),
        "unit_check": {"gauss_fill_write_size_KiB": gw[:1], "expected_KiB": 16 * 256 * 256 * 32 / 1024},
        "correction": "gfx950 FETCH_SIZE counts 64 B per 128-B request on wide coalesced reads "
                      "(MI355X_MICROARCH.md, HBM): read bytes = 2 x raw",
    }
    return res, out


def mfma(argv):
    d, pat, expected = argv[0], argv[1], float(argv[2])
    out = argv[3] if len(argv) > 3 else None
    busy, gui = load(d, "SQ_VALU_MFMA_BUSY_CYCLES"), load(d, "GRBM_GUI_ACTIVE")
    b = [v for k, v in busy if pat in k]
    g = [v for k, v in gui if pat in k]
    assert b and g and len(b) == len(g), "kernel %r not found (%d / %d rows)" % (pat, len(b), len(g))
    name = next(k for k, _ in busy if pat in k)
    simd_cycles = [x / 8.0 * 256 * 4 for x in g]          # cycles offered by the chip's 1024 SIMDs during the launch
    util = [bi / si for bi, si in zip(b, simd_cycles)]
    allb, allg = sum(v for _, v in busy), sum(v for _, v in gui)
    res = {
        "kernel": name.split("(")[0],
        "pattern": pat,
        "csrc_sha16": sha(),
        "population": "all %d launches of kernels whose name contains the pattern, in the profiled command" % len(b),
        "command": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -- python3 bench.py "
                   "--steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-sample-call",
        "sq_valu_mfma_busy_cycles_per_launch": sum(b) / len(b),
        "grbm_gui_active_per_launch_sum_over_8_xcd": sum(g) / len(g),
        "mfma_busy_frac": sum(b) / sum(simd_cycles),
        "mfma_busy_frac_min_max": [min(util), max(util)],
        "unit_check": {"expected_mfma_cycles_per_launch": expected, "counter_over_expected": (sum(b) / len(b)) / expected,
                       "meaning": "expected = MFMA instructions per launch x issue cycles per instruction, summed over all waves"},
        "whole_command": {"mfma_busy_frac_all_kernels": allb / (allg / 8.0 * 256 * 4), "kernels_counted": len(busy)},
        "normalisation": "busy cycles / (GRBM_GUI_ACTIVE / 8 XCDs x 256 CUs x 4 SIMDs); clock-independent (both sides count shader cycles)",
    }
    return res, out


def main():
    mode = sys.argv[1]
    res, out = (traffic if mode == "traffic" else mfma)(sys.argv[2:])
    if out:
        json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
