#!/bin/bash
# LDS-array counters of the shipped split tile (and the round-2 4-wave tile beside it): is the dominant kernel near the LDS limit?
cd /root/repo
R=/root/repo
cd /tmp && export TMPDIR=/tmp
CDX_TUNE=1 rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/r3lds -- python3 $R/tools/conv_bench.py --shape 16,256,256,128,0,128,3,1 --tiles 11,106 --gn --stats --rounds 1 --iters 3 > $R/gpurun_out/r3lds.log 2>&1
echo lds rc=$?
CDX_TUNE=1 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/r3wait -- python3 $R/tools/conv_bench.py --shape 16,256,256,128,0,128,3,1 --tiles 11,106 --gn --stats --rounds 1 --iters 3 > $R/gpurun_out/r3wait.log 2>&1
echo wait rc=$?
