#!/bin/bash
# whole GPU suite (range contract build + ADVICE fixes) and the default bench line
cd /root/repo
mkdir -p gpurun_out
timeout -k 10 1700 python -m pytest tests -m gpu -q > gpurun_out/r3d_suite.log 2>&1; echo "suite rc=$?"
tail -6 gpurun_out/r3d_suite.log
timeout -k 10 600 python bench.py > gpurun_out/r3d_bench.json 2> gpurun_out/r3d_bench.err; echo "bench rc=$?"
head -c 600 gpurun_out/r3d_bench.json
