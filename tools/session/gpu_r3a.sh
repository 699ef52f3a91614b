#!/bin/bash
# round 3, first GPU pass of the range-contract build: new range tests, kernel + e2e parity, isolated conv timings, bench
cd /root/repo
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_range_gpu.py -m gpu -x -q > gpurun_out/r3a_range.log 2>&1; echo "range rc=$?" 
tail -5 gpurun_out/r3a_range.log
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_e2e_gpu.py -m gpu -x -q > gpurun_out/r3a_kern.log 2>&1; echo "kern rc=$?"
tail -5 gpurun_out/r3a_kern.log
for sh in 16,256,256,128,0,128,3,1 16,256,256,256,0,128,1,1 16,128,128,128,0,128,3,2; do
  timeout -k 10 120 python tools/conv_bench.py --shape $sh --tiles 11 --gn --stats 2>&1 | grep -v amdgpu.ids
  timeout -k 10 120 python tools/conv_bench.py --shape $sh --tiles 11 --stats 2>&1 | grep -v amdgpu.ids
done
timeout -k 10 600 python bench.py --steps 20 --warmup 3 --details > gpurun_out/r3a_bench.json 2> gpurun_out/r3a_bench.err; echo "bench rc=$?"
cat gpurun_out/r3a_bench.json | head -c 1500
