#!/bin/bash
# final build of round 3: default bench line (+ details), bench contract tests, PMC passes (cfg2 / cfg4 / cfg5) and co-execution counters
cd /root/repo
mkdir -p gpurun_out
python bench.py --details > gpurun_out/r3u_bench_cfg2.json 2> gpurun_out/r3u_bench_cfg2.err; echo "cfg2 rc=$?"; head -c 300 gpurun_out/r3u_bench_cfg2.json; echo
timeout -k 10 600 python -m pytest tests/test_bench_contract_gpu.py tests/test_e2e_gpu.py tests/test_configs_gpu.py -m gpu -x -q 2>&1 | tail -3
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-/root/repo}
bash tools/gpu_prof.sh r3u 2>&1 | tail -2
bash tools/gpu_prof.sh r3u4 --config cfg4 2>&1 | tail -1
bash tools/gpu_prof.sh r3u5 --config cfg5 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
CDX_TUNE=1 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES --output-format csv -d /root/repo/gpurun_out/r3u_coexec -- python3 /root/repo/tools/conv_bench.py --shape 16,256,256,128,0,128,3,1 --tiles 11,106,102,107 --gn --stats --rounds 1 --iters 3 > /root/repo/gpurun_out/r3u_coexec.log 2>&1
echo coexec rc=$?
