#!/bin/bash
# round 4, call f: tile sequences of the wave-specialised kernels -- correctness at seq 2 / 4, then in-process A/B:
# tile 117 = ring depth 3, one tile per workgroup (call b's kernel), 113 / 114 / 115 / 116 = ring depth 2 with 1 / 2 / 4 / 8 tiles per workgroup,
# 11 = the library's pick; the 16-bit tile: abl 11 / 12 / 13 = 1 / 2 / 4 tiles, 0 = the pick
cd "$(dirname "$0")/../.." && . tools/session/r4lib.sh
step r4f_tests_seq 900 python -m pytest tests/test_boundary_gpu.py -q -x -k "sequences" --timeout 600
for shape in 16,256,256,128,0,128,3,1 16,256,256,128,128,128,3,1 16,128,128,256,0,256,3,1 16,64,64,256,0,256,3,1 8,512,512,192,0,192,3,1; do
    step "r4f_ab_${shape//,/_}" 400 python tools/conv_bench.py --shape $shape --tiles 117,113,114,115,116,11 --gn --stats --check --rounds 5
done
step r4f_ab16_a 200 python tools/conv16_bench.py --shape 16,256,256,128,0,128,3,1 --abl 11,12,13,0,11,12,13,0
step r4f_ab16_b 200 python tools/conv16_bench.py --shape 16,256,256,256,0,128,3,1 --abl 11,12,13,0,11,12,13,0
step r4f_ab16_c 200 python tools/conv16_bench.py --shape 16,128,128,256,0,256,3,1 --abl 11,12,13,0
step r4f_bench 500 python bench.py --no-parity-gate --cpu-budget 20
step r4f_bench_cfg5 300 python bench.py --config cfg5 --no-cpu-baseline
step r4f_bench_cfg4 400 python bench.py --config cfg4 --no-cpu-baseline --steps 10
