#!/bin/bash
# round-3 final build: whole GPU suite, bench lines of every config / dtype, kernel stats + PMC passes
cd /root/repo
mkdir -p gpurun_out
timeout -k 10 1700 python -m pytest tests -m gpu -q > gpurun_out/r3t_suite.log 2>&1; echo "suite rc=$?"; tail -3 gpurun_out/r3t_suite.log
python bench.py --details > gpurun_out/r3t_bench_cfg2.json 2> gpurun_out/r3t_bench_cfg2.err; echo "cfg2 rc=$?"
python bench.py --config cfg4 --steps 10 --warmup 2 > gpurun_out/r3t_bench_cfg4.json 2> gpurun_out/r3t_bench_cfg4.err
python bench.py --config cfg5 --steps 20 --warmup 2 > gpurun_out/r3t_bench_cfg5.json 2> gpurun_out/r3t_bench_cfg5.err
python bench.py --config cfg5 --dtype bf16 --steps 20 --warmup 2 --no-cpu-baseline --no-sample-call > gpurun_out/r3t_bench_cfg5_bf16.json 2> /dev/null
python bench.py --dtype fp16 --steps 50 --no-cpu-baseline --no-sample-call > gpurun_out/r3t_bench_cfg2_fp16.json 2> /dev/null
python bench.py --dtype bf16 --steps 50 --no-cpu-baseline --no-sample-call > gpurun_out/r3t_bench_cfg2_bf16.json 2> /dev/null
python bench.py --config cfg1 --steps 50 --no-cpu-baseline > gpurun_out/r3t_bench_cfg1.json 2> /dev/null
for f in gpurun_out/r3t_bench_*.json; do python -c "
import json,sys
d=json.load(open('$f')); print('$f', d['value'], d['unit'], d['ms_per_step'], 'ms/step', d['dtype'], d.get('roofline',{}).get('frac'), d.get('strict_f32',{}).get('images_per_s'))"; done
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-/root/repo}
bash tools/gpu_prof.sh r3t 2>&1 | tail -2
bash tools/gpu_prof.sh r3t4 --config cfg4 2>&1 | tail -1
bash tools/gpu_prof.sh r3t5 --config cfg5 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
CDX_TUNE=1 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES --output-format csv -d /root/repo/gpurun_out/r3t_coexec -- python3 /root/repo/tools/conv_bench.py --shape 16,256,256,128,0,128,3,1 --tiles 11,106,102,107 --gn --stats --rounds 1 --iters 3 > /root/repo/gpurun_out/r3t_coexec.log 2>&1
echo coexec rc=$?
