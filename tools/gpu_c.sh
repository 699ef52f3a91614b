#!/bin/bash
# GPU call C: ablations of the split tile + cfg4 / cfg5 bench lines + kernel-trace stats of the default bench.
set -o pipefail
mkdir -p gpurun_out
python tools/conv_bench.py --shape 16,256,256,128,0,128,3,1 --tiles 11,61,62,63,64,67,68,75,76,77 --gn --rounds 3 --iters 5 2>&1 | tee gpurun_out/r02_c_ablate.log &&
python tools/conv_bench.py --shape 16,256,256,128,128,128,3,1 --tiles 11,61,62,64,68,75,76 --gn --rounds 3 --iters 5 2>&1 | tee -a gpurun_out/r02_c_ablate.log &&
python bench.py --config cfg4 --steps 10 --warmup 2 --details > gpurun_out/r02_c_bench_cfg4.json 2> gpurun_out/r02_c_bench_cfg4.err && tail -c 1500 gpurun_out/r02_c_bench_cfg4.json &&
python bench.py --config cfg5 --steps 20 --warmup 2 --details > gpurun_out/r02_c_bench_cfg5.json 2> gpurun_out/r02_c_bench_cfg5.err && tail -c 1500 gpurun_out/r02_c_bench_cfg5.json &&
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02_c_prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-sample-call > $GRAFT_REPO_ROOT/gpurun_out/r02_c_prof.log 2>&1; tail -3 $GRAFT_REPO_ROOT/gpurun_out/r02_c_prof.log
