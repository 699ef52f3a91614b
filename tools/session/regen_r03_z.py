#!/usr/bin/env python3
"""Turn what tools/session/gpu_r3z.sh + gpu_r3z2.sh left under gpurun_out/ into the tracked profiles/r03_z_* files
(PMC digests stamped with the kernel-source hash, rocprofv3 kernel stats, bench lines, parity metrics)."""
import collections, csv, glob, json, os, shutil, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import bench
sha = bench.csrc_sha16(); print("sha", sha)
pat2 = 'Conv16Cfg<3, 1, 4, 4, 3, 0, 1, 1, 0, 1>'
pat5 = 'Conv16Cfg<3, 1, 4, 4, 3, 0, 0, 1, '
b2 = json.load(open(f'{R}/gpurun_out/r3z_bench_cfg2.json')); b5 = json.load(open(f'{R}/gpurun_out/r3z_bench_cfg5.json'))
assert b2['roofline']['csrc_sha16'] == sha, (b2['roofline']['csrc_sha16'], sha)
def run(args):
    r = subprocess.run([sys.executable, f'{R}/tools/pmc_digest.py'] + args, capture_output=True, text=True, cwd=R)
    d = json.loads(r.stdout) if r.returncode == 0 else None
    print(args[0], {k: d[k] for k in d if k in ('hbm_bytes_per_launch', 'mfma_busy_frac', 'csrc_sha16')} if d else r.stderr[-500:], d.get('unit_check') if d else '')
run(['traffic', f'{R}/gpurun_out/r3z_fetch', f'{R}/gpurun_out/r3z_write', pat2, f'{R}/profiles/r03_z_traffic.json', 'cfg2'])
run(['traffic', f'{R}/gpurun_out/r3z4_fetch', f'{R}/gpurun_out/r3z4_write', pat2, f'{R}/profiles/r03_z_traffic_cfg4.json', 'cfg4'])
run(['traffic', f'{R}/gpurun_out/r3z5_fetch', f'{R}/gpurun_out/r3z5_write', pat5, f'{R}/profiles/r03_z_traffic_cfg5.json', 'cfg5'])
run(['mfma', f'{R}/gpurun_out/r3z_mfma', pat2, str(b2['roofline']['mfma_cycles_per_launch_expected']), f'{R}/profiles/r03_z_mfma_busy.json'])
run(['mfma', f'{R}/gpurun_out/r3z5_mfma', pat5, str(b5['roofline']['mfma_cycles_per_launch_expected']), f'{R}/profiles/r03_z_mfma_busy_cfg5.json'])
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f'{R}/gpurun_out/r3z_coexec/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'conv16' in k or 'conv_wsp' in k:
            rows[k.split('(')[0].replace('void cdx::', '')][r['Counter_Name']].append(float(r['Counter_Value']))
names = {'conv16_ws_kernel<cdx::Conv16Cfg<3, 1, 4, 4, 3, 0, 1, 1, 0, 1>, 2>': 'shipped: wave-specialised 8-wave workgroup, 8 x 16-pixel tile (CDX_TILE_SPLIT; conv_bench tile 11)',
         'conv16_ws_kernel<cdx::Conv16Cfg<3, 1, 5, 4, 3, 0, 1, 1, 0, 1>, 2>': 'wave-specialised, 4 x 32-pixel tile (tuning tile 100)',
         'conv16_kernel<cdx::Conv16Cfg<3, 1, 5, 4, 3, 0, 1, 1, 0, 0>, 2>': 'round 2: homogeneous 4-wave workgroup, 4 x 32-pixel tile (tuning tile 106)',
         'conv_wsp_kernel<cdx::WspCfg<3, 0>, 2>': 'persistent wave-specialised (tuning tile 102)',
         'conv_wsp_kernel<cdx::WspCfg<3, 8>, 2>': 'persistent, producers at s_setprio 3 (tuning tile 107)'}
out = {"shape": "256x256, 128 -> 128, batch 16, 3x3, fused GroupNorm + SiLU + temb + residual + sums + amax (tools/conv_bench.py --gn --stats)",
       "command": "CDX_TUNE=1 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES -- python3 tools/conv_bench.py --shape 16,256,256,128,0,128,3,1 --tiles 11,100,106,102,107 --gn --stats --rounds 1 --iters 3",
       "normalisation": "mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 256 CUs x 4 SIMDs); coexec_over_busy = SQ_VALU_MFMA_COEXEC_CYCLES / SQ_VALU_MFMA_BUSY_CYCLES",
       "csrc_sha16": sha, "variants": {}}
for k, v in rows.items():
    m = {c: sum(x) / len(x) for c, x in v.items()}
    simd = m['GRBM_GUI_ACTIVE'] / 8 * 256 * 4
    out["variants"][names.get(k, k)] = {"kernel": k, "launches": len(v['GRBM_GUI_ACTIVE']),
        "mfma_busy_frac": round(m['SQ_VALU_MFMA_BUSY_CYCLES'] / simd, 4),
        "coexec_over_busy": round(m['SQ_VALU_MFMA_COEXEC_CYCLES'] / m['SQ_VALU_MFMA_BUSY_CYCLES'], 4),
        "valu_insts_per_mfma": round(m['SQ_INSTS_VALU'] / m['SQ_INSTS_MFMA'], 3),
        "gui_active_cycles_per_xcd": round(m['GRBM_GUI_ACTIVE'] / 8)}
json.dump(out, open(f'{R}/profiles/r03_z_coexec.json', 'w'), indent=1)
for k, v in out["variants"].items(): print(k[:70], v["mfma_busy_frac"], v["coexec_over_busy"], v["valu_insts_per_mfma"], v["gui_active_cycles_per_xcd"])
for tag, cfg in (('r3z', 'cfg2'), ('r3z4', 'cfg4'), ('r3z5', 'cfg5')):
    shutil.copy(glob.glob(f'{R}/gpurun_out/{tag}_stats/*/*kernel_stats.csv')[0], f'{R}/profiles/r03_z_kernel_stats_{cfg}.csv')
t = open(f'{R}/gpurun_out/r3z_bench_cfg2.err').read(); i = t.index('{\n "conv_variants"')
open(f'{R}/profiles/r03_z_conv_shapes.json', 'w').write(t[i:])
for c in ['cfg2', 'cfg1', 'cfg4', 'cfg5', 'cfg5_bf16', 'cfg2_fp16', 'cfg2_bf16']:
    shutil.copy(f'{R}/gpurun_out/r3z_bench_{c}.json', f'{R}/profiles/r03_z_bench_{c}.json')
shutil.copy(f'{R}/gpurun_out/metrics.jsonl', f'{R}/profiles/r03_z_parity_metrics.jsonl')
print({k: b2[k] for k in ('value', 'ms_per_step')}, b2['strict_f32']['images_per_s'], b2['cpu_baseline']['value'], b2['sample_call']['images_per_s'],
      {k: b2['roofline'][k] for k in ('achieved', 'frac', 'mfma_pipe_utilisation', 'avg_launch_ms', 'algorithmic_bytes_per_launch')})
for c in ['cfg1', 'cfg4', 'cfg5', 'cfg5_bf16', 'cfg2_fp16', 'cfg2_bf16']:
    d = json.load(open(f'{R}/gpurun_out/r3z_bench_{c}.json')); print(c, d['value'], d['ms_per_step'], d.get('roofline', {}).get('frac'))
