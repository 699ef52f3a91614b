#!/bin/bash
# 8 x 16-pixel wave-specialised tile adopted (split + 16-bit): whole GPU suite, bench lines, PMC passes, co-execution counters
cd /root/repo
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/r3z_suite.log 2>&1; echo "suite rc=$?"; tail -3 gpurun_out/r3z_suite.log
python bench.py --details > gpurun_out/r3z_bench_cfg2.json 2> gpurun_out/r3z_bench_cfg2.err; echo "cfg2 rc=$?"
python bench.py --config cfg4 --steps 10 --warmup 2 > gpurun_out/r3z_bench_cfg4.json 2> gpurun_out/r3z_bench_cfg4.err
python bench.py --config cfg5 --steps 20 --warmup 2 > gpurun_out/r3z_bench_cfg5.json 2> gpurun_out/r3z_bench_cfg5.err
python bench.py --config cfg5 --dtype bf16 --steps 20 --warmup 2 --no-cpu-baseline --no-sample-call > gpurun_out/r3z_bench_cfg5_bf16.json 2> /dev/null
python bench.py --dtype fp16 --steps 50 --no-cpu-baseline --no-sample-call > gpurun_out/r3z_bench_cfg2_fp16.json 2> /dev/null
python bench.py --dtype bf16 --steps 50 --no-cpu-baseline --no-sample-call > gpurun_out/r3z_bench_cfg2_bf16.json 2> /dev/null
python bench.py --config cfg1 --steps 50 --no-cpu-baseline > gpurun_out/r3z_bench_cfg1.json 2> /dev/null
for f in gpurun_out/r3z_bench_*.json; do python -c "
import json,sys
d=json.load(open('$f')); print('$f', d['value'], d['unit'], d['ms_per_step'], 'ms/step', d['dtype'], d.get('roofline',{}).get('frac'), d.get('strict_f32',{}).get('images_per_s'))"; done
