#!/bin/bash
# tile 12 (loads one block ahead): parity tests, ablations (tuning build), A/B against tile 8
cd /root/repo
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "conv" > gpurun_out/r3ac_suite.log 2>&1; echo "suite rc=$?"; tail -2 gpurun_out/r3ac_suite.log
for abl in 0 1 2 5 6; do
  echo "ABL $abl"; CDX_TUNE=1 CDX_GEMM_ABL=$abl timeout -k 10 200 python tools/conv_bench.py --shape 16,256,256,128,0,3,3,1 --tiles 12 --gn --rounds 3 2>&1 | tail -1
done
for shp in 16,256,256,128,0,3,3,1 8,512,512,192,0,3,3,1 16,256,256,256,0,3,3,1 16,256,256,64,0,3,3,1; do
  timeout -k 10 200 python tools/conv_bench.py --shape $shp --tiles 8,12 --gn --check --rounds 3 2>&1 | tail -3
done
