#!/usr/bin/env python3
"""What would a persistent / cooperative kernel for the low-resolution levels remove?  (second half; VERDICT r03
item 8: "measure the launch gaps it would remove ... and record the ceiling".)

Input: the kernel-trace CSV of `rocprofv3 --kernel-trace -- python3 bench.py --steps K --warmup W --no-roofline --no-sample-call
--no-strict-f32 --no-cpu-baseline` (one row per dispatch with Start_Timestamp / End_Timestamp in ns).  Dispatches are sorted by
start time; the gap behind dispatch i is start[i+1] - end[i] (the stream is in order: a negative gap cannot occur).  A dispatch is
"short" when it runs for less than --short microseconds: at cfg2 these are the 3x3 / 1x1 layers of the 16^2 and 8^2 levels, the
attention kernels, the GroupNorm finalizes, the timestep MLP, the range bookkeeping -- the launches a low-resolution megakernel or
a fused finalize would absorb.  Reported per forward (= per sampler step):
  * time inside short dispatches, and inside all dispatches;
  * idle time between dispatches, split by whether BOTH neighbours are short (the share a kernel that keeps the low-resolution
    levels resident could remove), one is, or none;
so the ceiling of row (f3)'s second half is "gaps between short dispatches" + whatever of "time inside short dispatches" a fused
form would run faster -- against the step time.

  python tools/launch_gaps.py gpurun_out/<dir>/**/*kernel_trace.csv --forwards 12 [--short 45]
"""
import argparse, csv, glob, json, sys

ap = argparse.ArgumentParser()
ap.add_argument("trace")
ap.add_argument("--forwards", type=int, required=True, help="UNet forwards in the traced command (warmup + steps)")
ap.add_argument("--short", type=float, default=45.0, help="dispatches shorter than this many microseconds count as low-resolution / bookkeeping")
ap.add_argument("--json", default=None)
a = ap.parse_args()
paths = glob.glob(a.trace, recursive=True)
assert len(paths) == 1, paths
rows = []
with open(paths[0]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# the timed forwards are the densest part of the trace: drop everything before the first UNet kernel of the first forward (weight
# packing, input generation) by starting at the first cdx_fill / timestep kernel
first = next(i for i, r in enumerate(rows) if "timestep_embedding" in r[2] or "fill_u32" in r[2])
rows = rows[first:]
short_ns = a.short * 1e3
dur = [e - s for s, e, _ in rows]
is_short = [d < short_ns for d in dur]
gaps = {"both_short": 0, "one_short": 0, "none_short": 0}
ngaps = dict(gaps)
big_idle = 0
for i in range(len(rows) - 1):
    g = rows[i + 1][0] - rows[i][1]
    if g < 0:
        continue          # (overlapping dispatches: independent kernels the hardware ran side by side)
    if g > 2e6:
        big_idle += g     # host-side pauses (between warm-up and the timed loop, the final sync): not launch gaps
        continue
    k = "both_short" if is_short[i] and is_short[i + 1] else "one_short" if is_short[i] or is_short[i + 1] else "none_short"
    gaps[k] += g
    ngaps[k] += 1
F = a.forwards
per = lambda ns: round(ns / F / 1e3, 1)      # noqa: E731  microseconds per forward
out = {"trace": paths[0].split("gpurun_out/")[-1], "forwards": F, "short_threshold_us": a.short,
       "dispatches_per_forward": round(len(rows) / F, 1), "short_dispatches_per_forward": round(sum(is_short) / F, 1),
       "us_per_forward": {"inside_all_dispatches": per(sum(dur)), "inside_short_dispatches": per(sum(d for d, s in zip(dur, is_short) if s)),
                          "gaps_between_short_dispatches": per(gaps["both_short"]), "gaps_with_one_short_neighbour": per(gaps["one_short"]),
                          "gaps_between_long_dispatches": per(gaps["none_short"])},
       "mean_gap_us": {k: round(gaps[k] / max(ngaps[k], 1) / 1e3, 2) for k in gaps},
       "gaps_counted_per_forward": {k: round(ngaps[k] / F, 1) for k in ngaps}}
tot = sum(dur) + sum(gaps.values())
out["share_of_step"] = {"gaps_between_short_dispatches": round(gaps["both_short"] / tot, 4), "all_gaps": round(sum(gaps.values()) / tot, 4),
                        "inside_short_dispatches": round(sum(d for d, s in zip(dur, is_short) if s) / tot, 4)}
print(json.dumps(out, indent=1))
if a.json:
    json.dump(out, open(a.json, "w"), indent=1)
