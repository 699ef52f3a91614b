#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py tests/test_fuzz_gpu.py tests/test_fp16_gpu.py -m gpu -x -q 2>&1 | tee gpurun_out/r02_f_tests.log | tail -6 &&
python tools/conv_bench.py --shape 16,256,256,128,0,128,3,1 --tiles 11,61,62,64,68,75 --gn --stats --rounds 3 --iters 5 2>&1 | tee gpurun_out/r02_f_convbench.log &&
python tools/conv_bench.py --shape 16,256,256,128,128,128,3,1 --tiles 11,7 --gn --stats --rounds 3 --iters 5 2>&1 | tee -a gpurun_out/r02_f_convbench.log &&
python tools/conv_bench.py --shape 16,64,64,256,0,256,3,1 --tiles 11 --gn --stats --rounds 3 --iters 5 2>&1 | tee -a gpurun_out/r02_f_convbench.log &&
python bench.py --details --steps 50 --no-cpu-baseline > gpurun_out/r02_f_bench.json 2> gpurun_out/r02_f_bench.err; tail -c 900 gpurun_out/r02_f_bench.json
