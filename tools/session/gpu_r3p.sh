#!/bin/bash
# round-3 profiles of the wave-specialised build: kernel stats + PMC passes (cfg2, cfg4 traffic, cfg5 busy), co-execution counters
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-/root/repo}
R=$GRAFT_REPO_ROOT
bash $R/tools/gpu_prof.sh r3p 2>&1 | tail -3
bash $R/tools/gpu_prof.sh r3p4 --config cfg4 2>&1 | tail -2
bash $R/tools/gpu_prof.sh r3p5 --config cfg5 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
CDX_TUNE=1 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/r3p_coexec -- python3 $R/tools/conv_bench.py --shape 16,256,256,128,0,128,3,1 --tiles 11,106,102,107 --gn --stats --rounds 1 --iters 3 > $R/gpurun_out/r3p_coexec.log 2>&1
echo coexec rc=$?
ls $R/gpurun_out | grep r3p | head -30; du -sh $R/gpurun_out
