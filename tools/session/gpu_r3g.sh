#!/bin/bash
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r3g_stats_cfg5 -- python3 /root/repo/bench.py --config cfg5 --steps 10 --warmup 2 --no-cpu-baseline --no-sample-call --no-roofline > /root/repo/gpurun_out/r3g_stats_cfg5.log 2>&1
echo "prof rc=$?"; grep -o '"value": [0-9.]*' /root/repo/gpurun_out/r3g_stats_cfg5.log | head -2
