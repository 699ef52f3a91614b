#!/bin/bash
# round 4, call b: software-pipelined operand reads in the MFMA waves -- in-process A/B (tile 110 = without, 111 = with, 11 = the
# library's pick) on the dominant shapes, the 16-bit tile (abl 8 = without), stamps of the new tile, then the parity suites
cd "$(dirname "$0")/../.." && . tools/session/r4lib.sh
for shape in 16,256,256,128,0,128,3,1 16,256,256,128,128,128,3,1 16,128,128,256,0,256,3,1 16,64,64,256,0,256,3,1 8,512,512,192,0,192,3,1; do
    step "r4b_ab_${shape//,/_}" 300 python tools/conv_bench.py --shape $shape --tiles 110,111,11 --gn --stats --check --rounds 5
done
step r4b_ab16_a 200 python tools/conv16_bench.py --shape 16,256,256,128,0,128,3,1 --abl 8,0,8,0
step r4b_ab16_b 200 python tools/conv16_bench.py --shape 16,256,256,256,0,128,3,1 --abl 8,0,8,0
step r4b_stamps 200 python tools/ws_stamps.py
step r4b_tests 1150 python -m pytest tests/test_boundary_gpu.py tests/test_range_gpu.py tests/test_kernels_gpu.py tests/test_fp16_gpu.py tests/test_bf16_gpu.py tests/test_e2e_gpu.py tests/test_bench_contract_gpu.py -q -x --timeout 900
step r4b_bench 500 python bench.py --no-parity-gate
step r4b_bench_cfg5 300 python bench.py --config cfg5 --no-cpu-baseline
step r4b_fuzz 400 python tools/fuzz_conv.py 300 501
