#!/bin/bash
# 8-wave split tile (variants 30 / 31) against the shipped split tile (11)
cd /root/repo
export CDX_TUNE=1
for sh in 16,256,256,128,0,128,3,1 16,128,128,256,0,256,3,1 16,256,256,256,128,128,3,1 2,64,64,64,0,192,3,1; do
  timeout -k 10 120 python tools/conv_bench.py --shape $sh --tiles 11,90,91 --check || exit 1
  timeout -k 10 120 python tools/conv_bench.py --shape $sh --tiles 11,90,91 --gn --stats --check || exit 1
done
