#!/bin/bash
# round 4, call c: XCD-contiguous workgroup order -- in-process A/B (tile 112 = plain order, 111 = remapped), FETCH_SIZE of both,
# the 16-bit tile (abl 9 = plain order), then bench lines
cd "$(dirname "$0")/../.." && . tools/session/r4lib.sh
R=$PWD
for shape in 16,256,256,128,0,128,3,1 16,256,256,128,128,128,3,1 16,128,128,256,0,256,3,1 8,512,512,192,0,192,3,1 8,256,256,384,0,384,3,1; do
    step "r4c_ab_${shape//,/_}" 300 python tools/conv_bench.py --shape $shape --tiles 112,111,11 --gn --stats --check --rounds 5
done
step r4c_ab16_a 200 python tools/conv16_bench.py --shape 16,256,256,128,0,128,3,1 --abl 9,0,9,0
step r4c_ab16_b 200 python tools/conv16_bench.py --shape 16,256,256,256,0,128,3,1 --abl 9,0,9,0
cd /tmp && export TMPDIR=/tmp
step_pmc() { ( cd /tmp; rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/$1 -- python3 $R/tools/conv_bench.py --shape $2 --tiles 112,111 --gn --stats --rounds 1 --iters 3 ) > $R/gpurun_out/$1.log 2>&1; python3 $R/tools/pmc_by_kernel.py $R/gpurun_out/$1 conv16 | tee -a $R/gpurun_out/r4c_fetch_summary.log; }
step_pmc r4c_fetch_a 16,256,256,128,0,128,3,1
step_pmc r4c_fetch_b 8,512,512,192,0,192,3,1
step_pmc r4c_fetch_c 8,256,256,384,0,384,3,1
cd $R
step r4c_bench 500 python bench.py --no-parity-gate --cpu-budget 20
step r4c_bench_cfg5 300 python bench.py --config cfg5 --no-cpu-baseline
step r4c_bench_cfg4 400 python bench.py --config cfg4 --no-cpu-baseline --steps 10
step r4c_tests 900 python -m pytest tests/test_kernels_gpu.py tests/test_fp16_gpu.py tests/test_configs_gpu.py -q -x --timeout 900
