#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py tests/test_fuzz_gpu.py tests/test_fp16_gpu.py tests/test_bf16_gpu.py -m gpu -x -q 2>&1 | tee gpurun_out/r02_l_tests.log | tail -5 &&
python tools/conv_bench.py --shape 16,256,256,128,0,128,3,1 --tiles 11,61,62,64,75 --gn --stats --rounds 3 --iters 5 2>&1 | tee gpurun_out/r02_l_convbench.log &&
python bench.py --details --steps 50 --no-cpu-baseline > gpurun_out/r02_l_bench.json 2> gpurun_out/r02_l_bench.err; python -c "
import json; d=json.load(open('gpurun_out/r02_l_bench.json')); print('cfg2', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['algorithmic_tflops'])"
