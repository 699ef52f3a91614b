#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_fp16_gpu.py tests/test_bf16_gpu.py tests/test_configs_gpu.py -m gpu -x -q 2>&1 | tee gpurun_out/r02_k_tests.log | tail -5 &&
python bench.py --config cfg5 --steps 20 --warmup 2 --no-cpu-baseline > gpurun_out/r02_k_bench_cfg5.json 2> gpurun_out/r02_k_bench_cfg5.err && python -c "
import json; d=json.load(open('gpurun_out/r02_k_bench_cfg5.json')); print('cfg5', d['value'], d['ms_per_step'], d['roofline']['frac'])" &&
python bench.py --dtype fp16 --steps 50 --no-cpu-baseline --no-sample-call > gpurun_out/r02_k_bench_cfg2_fp16.json 2>/dev/null && python -c "
import json; d=json.load(open('gpurun_out/r02_k_bench_cfg2_fp16.json')); print('cfg2 fp16', d['value'], d['ms_per_step'], d['roofline']['frac'])"
