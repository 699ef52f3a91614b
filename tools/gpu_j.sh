#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_fp16_gpu.py tests/test_bf16_gpu.py tests/test_kernels_gpu.py -m gpu -x -q 2>&1 | tee gpurun_out/r02_j_tests.log | tail -5 &&
python bench.py --config cfg5 --steps 20 --warmup 2 --no-cpu-baseline > gpurun_out/r02_j_bench_cfg5.json 2> gpurun_out/r02_j_bench_cfg5.err && tail -c 300 gpurun_out/r02_j_bench_cfg5.json &&
python bench.py --steps 50 --no-cpu-baseline --no-sample-call > gpurun_out/r02_j_bench.json 2> gpurun_out/r02_j_bench.err; tail -c 200 gpurun_out/r02_j_bench.json
