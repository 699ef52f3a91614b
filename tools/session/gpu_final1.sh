#!/bin/bash
# Final session of a round: the whole GPU suite, the bench lines of every config / dtype, the profile set.
set -o pipefail
R=$GRAFT_REPO_ROOT
TAG=$1
mkdir -p gpurun_out; rm -f gpurun_out/metrics.jsonl
python -m pytest tests -m gpu -x -q --durations=8 2>&1 | tee gpurun_out/${TAG}_tests.log | tail -14 &&
python bench.py --details > gpurun_out/${TAG}_bench_cfg2.json 2> gpurun_out/${TAG}_bench_cfg2.err
echo "part1 rc=$?"
