#!/bin/bash
# round 4, call j: wide (LDS-transposed, 16-byte) epilogue of the float32 wave-specialised tile -- parity first, then in-process A/B
# (tile 123 = narrow epilogue, 11 = shipped), bench lines
cd "$(dirname "$0")/../.." && . tools/session/r4lib.sh
step r4j_tests 1000 python -m pytest tests/test_kernels_gpu.py tests/test_boundary_gpu.py tests/test_range_gpu.py tests/test_e2e_gpu.py -q -x --timeout 900
for shape in 16,256,256,128,0,128,3,1 16,256,256,128,128,128,3,1 16,128,128,256,0,256,3,1 16,64,64,256,0,256,3,1 8,512,512,192,0,192,3,1; do
    step "r4j_ab_${shape//,/_}" 300 python tools/conv_bench.py --shape $shape --tiles 123,11,123,11 --gn --stats --check --rounds 5
done
step r4j_bench 500 python bench.py --no-parity-gate --cpu-budget 20
step r4j_bench_cfg4 400 python bench.py --config cfg4 --no-cpu-baseline --steps 10
