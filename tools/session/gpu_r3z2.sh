#!/bin/bash
# PMC passes + co-execution counters of the final build
cd /root/repo
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-/root/repo}
bash tools/gpu_prof.sh r3z 2>&1 | tail -1
bash tools/gpu_prof.sh r3z4 --config cfg4 2>&1 | tail -1
bash tools/gpu_prof.sh r3z5 --config cfg5 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
CDX_TUNE=1 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES --output-format csv -d /root/repo/gpurun_out/r3z_coexec -- python3 /root/repo/tools/conv_bench.py --shape 16,256,256,128,0,128,3,1 --tiles 11,100,106,102,107 --gn --stats --rounds 1 --iters 3 > /root/repo/gpurun_out/r3z_coexec.log 2>&1
echo coexec rc=$?
