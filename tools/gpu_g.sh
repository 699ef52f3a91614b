#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_bf16_gpu.py tests/test_fp16_gpu.py tests/test_fuzz_gpu.py -m gpu -x -q 2>&1 | tee gpurun_out/r02_g_tests.log | tail -12 &&
python bench.py --config cfg5 --steps 20 --warmup 2 --no-cpu-baseline > gpurun_out/r02_g_bench_cfg5.json 2> gpurun_out/r02_g_bench_cfg5.err && tail -c 700 gpurun_out/r02_g_bench_cfg5.json
