#!/bin/bash
# A/B: shipped libcdx.so (built from HEAD) against libcdx_tune.so (built from the working tree), same process order per shape
cd /root/repo
for sh in 16,256,256,128,0,128,3,1 16,128,128,256,0,256,3,1 16,256,256,256,128,128,3,1 16,64,64,256,0,256,3,1 16,256,256,256,0,128,1,1; do
  for rep in 1 2; do
    timeout -k 10 120 python tools/conv_bench.py --shape $sh --tiles 11 --gn --stats | grep -v amdgpu.ids | sed 's/^/old /' || exit 1
    CDX_TUNE=1 timeout -k 10 120 python tools/conv_bench.py --shape $sh --tiles 11 --gn --stats | grep -v amdgpu.ids | sed 's/^/new /' || exit 1
  done
done
CDX_TUNE=1 timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_configs_gpu.py -m gpu -x -q 2>&1 | tail -5
