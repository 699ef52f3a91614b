#!/bin/bash
# bench lines of the final build WITH the traffic digests of its own source hash in profiles/ (the lines of gpu_r3z.sh were taken before them)
cd /root/repo
mkdir -p gpurun_out
python bench.py --details > gpurun_out/r3z_bench_cfg2.json 2> gpurun_out/r3z_bench_cfg2.err; echo "cfg2 rc=$?"
python bench.py --config cfg4 --steps 10 --warmup 2 > gpurun_out/r3z_bench_cfg4.json 2> gpurun_out/r3z_bench_cfg4.err
python bench.py --config cfg5 --steps 20 --warmup 2 > gpurun_out/r3z_bench_cfg5.json 2> gpurun_out/r3z_bench_cfg5.err
for f in gpurun_out/r3z_bench_cfg2.json gpurun_out/r3z_bench_cfg4.json gpurun_out/r3z_bench_cfg5.json; do python -c "
import json,sys
d=json.load(open('$f')); print('$f', d['value'], d['unit'], d['ms_per_step'], 'ms/step', d['dtype'], d.get('roofline',{}).get('frac'), d.get('roofline',{}).get('traffic'), d.get('strict_f32',{}).get('images_per_s'))"; done
