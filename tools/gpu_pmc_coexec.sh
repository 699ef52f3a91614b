#!/bin/bash
# matrix / vector co-execution counters of the dominant split-tile shape: two workgroups per CU (tile 11) and one (tile 160)
set -o pipefail
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export CDX_TUNE=1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/coexec -- python3 $R/tools/conv_bench.py --shape 16,256,256,128,0,128,3,1 --tiles 11,160,62,162 --gn --stats --rounds 1 --iters 3 > $R/gpurun_out/coexec.log 2>&1
echo rc=$?
