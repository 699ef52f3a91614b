#!/bin/bash
# same-box kernel traces of the default bench: round-2 tree (worktree _old_r02) and the current tree
cd /tmp && export TMPDIR=/tmp
cd /root/repo/_old_r02 && rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r3m_old_cfg2 -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-sample-call --no-roofline > /root/repo/gpurun_out/r3m_old.log 2>&1

cd /root/repo && rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r3m_new_cfg2 -- python3 /root/repo/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-sample-call --no-roofline --no-strict-f32 > /root/repo/gpurun_out/r3m_new.log 2>&1
grep -o '"ms_per_step": [0-9.]*' /root/repo/gpurun_out/r3m_old.log /root/repo/gpurun_out/r3m_new.log
