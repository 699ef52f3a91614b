#!/bin/bash
# fp16-storage dominant shape: timing ablations, then SQ instruction counters
set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 200 python tools/conv16_bench.py --shape 16,256,256,128,0,128,3,1 > gpurun_out/f16abl.log 2>&1 || exit 1
timeout -k 10 200 python tools/conv16_bench.py --shape 16,256,256,128,0,128,3,1 --plain >> gpurun_out/f16abl.log 2>&1 || exit 1
timeout -k 10 200 python tools/conv16_bench.py --shape 16,128,128,256,0,256,3,1 --abl 0,1,2,4,7 >> gpurun_out/f16abl.log 2>&1 || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVES --output-format csv -d $R/gpurun_out/f16sq -- python3 $R/tools/conv16_bench.py --shape 16,256,256,128,0,128,3,1 --abl 0 > $R/gpurun_out/f16sq.log 2>&1
echo rc=$?
