#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py tests/test_fuzz_gpu.py tests/test_fp16_gpu.py tests/test_configs_gpu.py -m gpu -x -q 2>&1 | tee gpurun_out/r02_i_tests.log | tail -6 &&
python bench.py --config cfg4 --steps 10 --warmup 2 --no-cpu-baseline --no-sample-call > gpurun_out/r02_i_bench_cfg4.json 2> gpurun_out/r02_i_bench_cfg4.err && tail -c 500 gpurun_out/r02_i_bench_cfg4.json &&
python bench.py --steps 50 --no-cpu-baseline > gpurun_out/r02_i_bench.json 2> gpurun_out/r02_i_bench.err; tail -c 300 gpurun_out/r02_i_bench.json
