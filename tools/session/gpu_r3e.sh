#!/bin/bash
# wave-specialised split tiles (tuning build: 100 = 8-wave WS, 102 = persistent WS) against the shipped tile 11, in-process A/B
cd /root/repo
for sh in 16,256,256,128,0,128,3,1 16,256,256,256,0,128,3,1 16,128,128,256,0,256,3,1 16,64,64,256,0,256,3,1 2,40,72,64,32,128,3,1; do
  timeout -k 10 200 python tools/conv_bench.py --shape $sh --tiles 11,100,102 --gn --stats --check --rounds 7 --iters 10 2>&1 | grep -v amdgpu.ids
done
timeout -k 10 200 python tools/conv_bench.py --shape 16,256,256,128,0,128,3,1 --tiles 11,100,102 --stats --check --rounds 5 --iters 10 2>&1 | grep -v amdgpu.ids
