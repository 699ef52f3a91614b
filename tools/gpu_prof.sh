#!/bin/bash
# Profiles of the default bench command for the build in the tree: kernel-trace stats + three PMC passes (separate runs).
# usage: bash tools/gpu_prof.sh <tag> [bench args...]     -> gpurun_out/<tag>_{stats,fetch,write,mfma}/
set -o pipefail
TAG=$1; shift
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-sample-call --no-strict-f32 $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-sample-call --no-strict-f32 --no-live-traffic "$@" > $R/gpurun_out/${TAG}_stats.log 2>&1 &&
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${TAG}_fetch -- python3 $R/bench.py $ARGS > $R/gpurun_out/${TAG}_fetch.log 2>&1 &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${TAG}_write -- python3 $R/bench.py $ARGS > $R/gpurun_out/${TAG}_write.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/${TAG}_mfma -- python3 $R/bench.py $ARGS > $R/gpurun_out/${TAG}_mfma.log 2>&1
echo "profiles done rc=$?"; ls $R/gpurun_out/${TAG}_*/ | head -20; tail -2 $R/gpurun_out/${TAG}_stats.log
