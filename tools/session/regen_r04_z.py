#!/usr/bin/env python3
"""Turn what tools/session/gpu_r4z2.sh left under gpurun_out/ into the tracked profiles/r04_z_* files (bench lines, rocprofv3 kernel
stats, PMC digests stamped with the kernel-source hash)."""
import glob, json, os, shutil, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import bench
sha = bench.csrc_sha16(); print("sha", sha)
names = ["cfg2", "cfg4", "cfg5", "cfg5_bf16", "cfg2_fp16", "cfg2_bf16", "cfg1"]
B = {}
for n in names:
    ln = open(f"{R}/gpurun_out/r4z_bench_{n}.json").read()
    B[n] = json.loads(ln)
    open(f"{R}/profiles/r04_z_bench_{n}.json", "w").write(ln)
assert B["cfg2"]["roofline"]["csrc_sha16"] == sha, (B["cfg2"]["roofline"]["csrc_sha16"], sha)
pat2, pat5 = B["cfg2"]["roofline"]["pmc_pattern"], B["cfg5"]["roofline"]["pmc_pattern"]
def run(args):
    r = subprocess.run([sys.executable, f"{R}/tools/pmc_digest.py"] + args, capture_output=True, text=True, cwd=R)
    if r.returncode:
        print("ERR", args[0], r.stderr[-800:]); return
    d = json.loads(r.stdout)
    print(args[0], {k: d[k] for k in d if k in ("hbm_bytes_per_launch", "mfma_busy_frac", "unit_check")})
run(["traffic", f"{R}/gpurun_out/r4z_fetch", f"{R}/gpurun_out/r4z_write", pat2, f"{R}/profiles/r04_z_traffic.json", "cfg2"])
run(["traffic", f"{R}/gpurun_out/r4z4_fetch", f"{R}/gpurun_out/r4z4_write", pat2, f"{R}/profiles/r04_z_traffic_cfg4.json", "cfg4"])
run(["traffic", f"{R}/gpurun_out/r4z5_fetch", f"{R}/gpurun_out/r4z5_write", pat5, f"{R}/profiles/r04_z_traffic_cfg5.json", "cfg5"])
run(["mfma", f"{R}/gpurun_out/r4z_mfma", pat2, str(B["cfg2"]["roofline"]["mfma_cycles_per_launch_expected"]), f"{R}/profiles/r04_z_mfma_busy.json"])
run(["mfma", f"{R}/gpurun_out/r4z5_mfma", pat5, str(B["cfg5"]["roofline"]["mfma_cycles_per_launch_expected"]), f"{R}/profiles/r04_z_mfma_busy_cfg5.json"])
for tag, cfg in (("r4z", "cfg2"), ("r4z4", "cfg4"), ("r4z5", "cfg5")):
    f = sorted(glob.glob(f"{R}/gpurun_out/{tag}_stats/*/*kernel_stats.csv"), key=os.path.getmtime)[-1]
    shutil.copy(f, f"{R}/profiles/r04_z_kernel_stats_{cfg}.csv")
if os.path.exists(f"{R}/gpurun_out/r4z_conv_shapes.json"):
    shutil.copy(f"{R}/gpurun_out/r4z_conv_shapes.json", f"{R}/profiles/r04_z_conv_shapes.json")
for n in names:
    d = B[n]
    print(n, d["value"], d["ms_per_step"], d.get("roofline", {}).get("frac"), d.get("roofline", {}).get("avg_launch_ms"), (d.get("parity") or {}).get("psnr_hip_vs_oracle_db"),
          (d.get("strict_f32") or {}).get("images_per_s"), (d.get("cpu_baseline") or {}).get("value"), (d.get("sample_call") or {}).get("images_per_s"))
