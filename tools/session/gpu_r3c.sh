#!/bin/bash
# whole GPU suite + kernel-trace stats of the default bench
cd /root/repo
mkdir -p gpurun_out
timeout -k 10 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r3c_suite.log 2>&1; echo "suite rc=$?"
tail -4 gpurun_out/r3c_suite.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r3c_stats -- python3 /root/repo/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-sample-call > /root/repo/gpurun_out/r3c_stats.log 2>&1
echo "prof rc=$?"; tail -c 600 /root/repo/gpurun_out/r3c_stats.log
