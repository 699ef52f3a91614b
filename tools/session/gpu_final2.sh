#!/bin/bash
# Final session, part 2: bench lines of the other configs / dtypes.
set -o pipefail
TAG=$1
mkdir -p gpurun_out
python bench.py --config cfg4 --steps 10 --warmup 2 > gpurun_out/${TAG}_bench_cfg4.json 2> gpurun_out/${TAG}_bench_cfg4.err &&
python bench.py --config cfg5 --steps 20 --warmup 2 > gpurun_out/${TAG}_bench_cfg5.json 2> gpurun_out/${TAG}_bench_cfg5.err &&
python bench.py --config cfg5 --dtype bf16 --steps 20 --warmup 2 --no-cpu-baseline --no-sample-call > gpurun_out/${TAG}_bench_cfg5_bf16.json 2> /dev/null &&
python bench.py --dtype fp16 --steps 50 --no-cpu-baseline --no-sample-call > gpurun_out/${TAG}_bench_cfg2_fp16.json 2> /dev/null &&
python bench.py --dtype bf16 --steps 50 --no-cpu-baseline --no-sample-call > gpurun_out/${TAG}_bench_cfg2_bf16.json 2> /dev/null &&
python bench.py --no-split --steps 30 --no-cpu-baseline --no-sample-call > gpurun_out/${TAG}_bench_cfg2_f32mfma.json 2> /dev/null &&
python bench.py --config cfg1 --steps 50 --no-cpu-baseline > gpurun_out/${TAG}_bench_cfg1.json 2> /dev/null
echo "bench rc=$?"
for f in gpurun_out/${TAG}_bench_*.json; do python -c "
import json,sys
d=json.load(open('$f')); print('$f', d['value'], d['unit'], d['ms_per_step'], 'ms/step', d['dtype'], d.get('roofline',{}).get('frac'))"; done
