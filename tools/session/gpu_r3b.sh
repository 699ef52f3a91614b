#!/bin/bash
cd /root/repo
for sh in 16,8,8,512,0,512,3,1 16,32,32,256,0,256,3,1 16,16,16,512,0,512,3,1; do
  for fl in "--gn --stats" "--gn --stats --no-amax" "--gn" "--stats" ""; do
    timeout -k 10 120 python tools/conv_bench.py --shape $sh --tiles 11 $fl --rounds 7 --iters 20 2>&1 | grep -v amdgpu.ids | sed "s/^/[$fl] /"
  done
done
