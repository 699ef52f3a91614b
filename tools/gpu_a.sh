#!/bin/bash
# GPU call A: full GPU test suite + the default bench line.
set -o pipefail
mkdir -p gpurun_out
rm -f gpurun_out/metrics.jsonl
python -m pytest tests -m gpu -x -q --durations=15 2>&1 | tee gpurun_out/r02_tests.log | tail -40 &&
python bench.py --details > gpurun_out/r02_bench_cfg2.json 2> gpurun_out/r02_bench_cfg2.err; tail -c 3000 gpurun_out/r02_bench_cfg2.json
