#!/bin/bash
# final build of round 3 (flat amax walk, conv_in padded to 8 channels): parity subset, bench lines, PMC passes, co-execution counters
cd /root/repo
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_e2e_gpu.py tests/test_configs_gpu.py tests/test_range_gpu.py tests/test_kernels_gpu.py tests/test_context_gpu.py -m gpu -x -q 2>&1 | tail -3
python bench.py --details > gpurun_out/r3y_bench_cfg2.json 2> gpurun_out/r3y_bench_cfg2.err; echo "cfg2 rc=$?"
python bench.py --config cfg4 --steps 10 --warmup 2 > gpurun_out/r3y_bench_cfg4.json 2> gpurun_out/r3y_bench_cfg4.err
python bench.py --config cfg1 --steps 50 --no-cpu-baseline > gpurun_out/r3y_bench_cfg1.json 2> /dev/null
for f in gpurun_out/r3y_bench_*.json; do python -c "
import json,sys
d=json.load(open('$f')); print('$f', d['value'], d['unit'], d['ms_per_step'], 'ms/step', d['dtype'], d.get('roofline',{}).get('frac'), d.get('roofline',{}).get('traffic'), d.get('strict_f32',{}).get('images_per_s'))"; done
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-/root/repo}
bash tools/gpu_prof.sh r3y 2>&1 | tail -1
bash tools/gpu_prof.sh r3y4 --config cfg4 2>&1 | tail -1
bash tools/gpu_prof.sh r3y5 --config cfg5 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
CDX_TUNE=1 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES --output-format csv -d /root/repo/gpurun_out/r3y_coexec -- python3 /root/repo/tools/conv_bench.py --shape 16,256,256,128,0,128,3,1 --tiles 11,106,102,107 --gn --stats --rounds 1 --iters 3 > /root/repo/gpurun_out/r3y_coexec.log 2>&1
echo coexec rc=$?
