#!/bin/bash
# SQ wave-state counters of the dominant split-tile shape (in-process bench of one conv shape).
set -o pipefail
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/r02_sq1 -- python3 $R/tools/conv_bench.py --shape 16,256,256,128,0,128,3,1 --tiles 11 --gn --stats --rounds 1 --iters 3 > $R/gpurun_out/r02_sq1.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC --output-format csv -d $R/gpurun_out/r02_sq2 -- python3 $R/tools/conv_bench.py --shape 16,256,256,128,0,128,3,1 --tiles 11 --gn --stats --rounds 1 --iters 3 > $R/gpurun_out/r02_sq2.log 2>&1
echo rc=$?; tail -2 $R/gpurun_out/r02_sq2.log
