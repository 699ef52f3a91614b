#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
rm -f gpurun_out/metrics.jsonl
python -m pytest tests -m gpu -x -q 2>&1 | tee gpurun_out/r02_d_tests.log | tail -8 &&
python tools/conv_bench.py --shape 16,16,16,512,0,512,3,1 --tiles 11,6 --gn --rounds 3 --iters 10 2>&1 | tee gpurun_out/r02_d_convbench.log &&
python tools/conv_bench.py --shape 16,16,16,512,0,1536,1,1 --tiles 11,1 --rounds 3 --iters 10 2>&1 | tee -a gpurun_out/r02_d_convbench.log &&
python tools/conv_bench.py --shape 16,256,256,128,0,128,3,2 --tiles 11,3 --rounds 3 --iters 5 2>&1 | tee -a gpurun_out/r02_d_convbench.log &&
python tools/conv_bench.py --shape 16,64,64,256,0,256,3,2 --tiles 11,3 --rounds 3 --iters 10 2>&1 | tee -a gpurun_out/r02_d_convbench.log &&
python bench.py --details --steps 50 > gpurun_out/r02_d_bench.json 2> gpurun_out/r02_d_bench.err; tail -c 1800 gpurun_out/r02_d_bench.json
