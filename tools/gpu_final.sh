#!/bin/bash
# Final session of a round: the whole GPU suite, the bench lines of every config / dtype, the profile set.
set -o pipefail
R=$GRAFT_REPO_ROOT
TAG=$1
mkdir -p gpurun_out; rm -f gpurun_out/metrics.jsonl
python -m pytest tests -m gpu -x -q --durations=8 2>&1 | tee gpurun_out/${TAG}_tests.log | tail -14 &&
python bench.py --details > gpurun_out/${TAG}_bench_cfg2.json 2> gpurun_out/${TAG}_bench_cfg2.err &&
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
