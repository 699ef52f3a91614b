#!/bin/bash
# Final profile set: cfg2 (stats + FETCH + WRITE + MFMA busy), cfg5 (stats + FETCH + WRITE), cfg4 (stats), + bench lines.
set -o pipefail
R=$GRAFT_REPO_ROOT
TAG=$1
bash $R/tools/gpu_prof.sh ${TAG}_cfg2 &&
cd /tmp && export TMPDIR=/tmp &&
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_cfg5_stats -- python3 $R/bench.py --config cfg5 --steps 10 --warmup 2 --no-cpu-baseline --no-sample-call > $R/gpurun_out/${TAG}_cfg5_stats.log 2>&1 &&
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${TAG}_cfg5_fetch -- python3 $R/bench.py --config cfg5 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-sample-call > $R/gpurun_out/${TAG}_cfg5_fetch.log 2>&1 &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${TAG}_cfg5_write -- python3 $R/bench.py --config cfg5 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-sample-call > $R/gpurun_out/${TAG}_cfg5_write.log 2>&1 &&
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_cfg4_stats -- python3 $R/bench.py --config cfg4 --steps 4 --warmup 1 --no-cpu-baseline --no-sample-call > $R/gpurun_out/${TAG}_cfg4_stats.log 2>&1
echo "rc=$?"
cd $R && python bench.py --details > gpurun_out/${TAG}_bench_cfg2.json 2> gpurun_out/${TAG}_bench_cfg2.err; tail -c 400 gpurun_out/${TAG}_bench_cfg2.json
