#!/bin/bash
# round 4, call d: in-kernel clock + phase stamps of the shipped tile; wave-state / MFMA-busy counters of the tile with and without
# pipelined operand reads (tile 11 vs 110), same command for both; launch gaps of the whole step (kernel trace of bench.py)
cd "$(dirname "$0")/../.." && . tools/session/r4lib.sh
R=$PWD
step r4d_stamps 200 python tools/ws_stamps.py
step r4d_stamps_mfma_bound 200 python tools/ws_stamps.py --tile 109
cd /tmp && export TMPDIR=/tmp
export CDX_TUNE=1
CB="python3 $R/tools/conv_bench.py --shape 16,256,256,128,0,128,3,1 --tiles 11,110 --gn --stats --rounds 1 --iters 3"
( timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/r4d_wait -- $CB ) > $R/gpurun_out/r4d_wait.log 2>&1; echo "wait rc=$?"
( timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/r4d_busy -- $CB ) > $R/gpurun_out/r4d_busy.log 2>&1; echo "busy rc=$?"
( timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/r4d_lds -- $CB ) > $R/gpurun_out/r4d_lds.log 2>&1; echo "lds rc=$?"
unset CDX_TUNE
( timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r4d_trace -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-sample-call --no-strict-f32 --no-roofline ) > $R/gpurun_out/r4d_trace.log 2>&1; echo "trace rc=$?"
cd $R
python3 tools/pmc_by_kernel.py gpurun_out/r4d_wait conv16 > gpurun_out/r4d_counters.log
python3 tools/pmc_by_kernel.py gpurun_out/r4d_busy conv16 >> gpurun_out/r4d_counters.log
python3 tools/pmc_by_kernel.py gpurun_out/r4d_lds conv16 >> gpurun_out/r4d_counters.log
cat gpurun_out/r4d_counters.log
python3 tools/launch_gaps.py "gpurun_out/r4d_trace/**/*kernel_trace.csv" --forwards 12 --json gpurun_out/r4d_launch_gaps.json
rm -rf gpurun_out/r4d_wait gpurun_out/r4d_busy gpurun_out/r4d_lds
find gpurun_out/r4d_trace -name "*.csv" ! -name "*kernel_trace.csv" -delete
