#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python tools/conv_bench.py --shape 16,256,256,128,0,128,3,1 --tiles 11,61,80,81,82,62,83,84,64,75 --gn --stats --rounds 3 --iters 5 2>&1 | tee gpurun_out/r02_e_ablate.log
