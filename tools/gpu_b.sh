#!/bin/bash
# GPU call B: first run of the split-fp16 tile: kernel parity, in-process A/B vs Winograd / direct, bench both ways.
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py tests/test_fuzz_gpu.py -m gpu -x -q 2>&1 | tee gpurun_out/r02_b_tests.log | tail -15 &&
for sh in 16,256,256,128,0,128,3,1 16,256,256,128,128,128,3,1 16,64,64,256,0,256,3,1 16,32,32,256,0,256,3,1 16,256,256,128,128,128,1,1; do
  python tools/conv_bench.py --shape $sh --tiles 11,7,0 --gn --rounds 3 --iters 5 2>&1 | tee -a gpurun_out/r02_b_convbench.log
done &&
python bench.py --details --steps 50 > gpurun_out/r02_b_bench_split.json 2> gpurun_out/r02_b_bench_split.err; tail -c 2500 gpurun_out/r02_b_bench_split.json
