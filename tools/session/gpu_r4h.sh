#!/bin/bash
# round 4, call h: two-phase residual fetch of the wave-specialised tiles -- stamps, in-process A/B (tile 118 = one-phase init, 11 = shipped),
# the 16-bit tile (abl 14 = one-phase), parity suites, bench lines
cd "$(dirname "$0")/../.." && . tools/session/r4lib.sh
step r4h_stamps 200 python tools/ws_stamps.py
step r4h_stamps16 200 python tools/ws_stamps.py --fp16
for shape in 16,256,256,128,0,128,3,1 16,256,256,128,128,128,3,1 16,128,128,256,0,256,3,1 16,64,64,256,0,256,3,1 8,512,512,192,0,192,3,1; do
    step "r4h_ab_${shape//,/_}" 300 python tools/conv_bench.py --shape $shape --tiles 118,11,118,11 --gn --stats --check --rounds 5
done
step r4h_ab16_a 200 python tools/conv16_bench.py --shape 16,256,256,128,0,128,3,1 --abl 14,0,14,0
step r4h_ab16_b 200 python tools/conv16_bench.py --shape 16,256,256,256,0,128,3,1 --abl 14,0,14,0
step r4h_tests 900 python -m pytest tests/test_kernels_gpu.py tests/test_fp16_gpu.py tests/test_bf16_gpu.py tests/test_boundary_gpu.py tests/test_range_gpu.py -q -x --timeout 900
step r4h_bench 500 python bench.py --no-parity-gate --cpu-budget 20
step r4h_bench_cfg5 300 python bench.py --config cfg5 --no-cpu-baseline
