#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py tests/test_fuzz_gpu.py tests/test_e2e_gpu.py -m gpu -x -q 2>&1 | tee gpurun_out/r02_h_tests.log | tail -6 &&
python bench.py --details --steps 50 --no-cpu-baseline > gpurun_out/r02_h_bench.json 2> gpurun_out/r02_h_bench.err; tail -c 600 gpurun_out/r02_h_bench.json
