#!/bin/bash
cd /root/repo
for sh in 16,256,256,128,0,128,3,1 16,256,256,256,0,128,1,1 16,256,256,128,0,128,3,2; do
 for fl in "--gn --stats" "--stats"; do
  (cd _old_r02 && python tools/conv_bench.py --shape $sh --tiles 11 $fl --rounds 7 --iters 10 2>&1 | grep -v amdgpu.ids | sed 's/^/r02 /')
  python tools/conv_bench.py --shape $sh --tiles 11,106 $fl --rounds 7 --iters 10 2>&1 | grep -v amdgpu.ids | sed 's/^/now /'
 done
done
