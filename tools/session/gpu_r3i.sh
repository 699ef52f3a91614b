#!/bin/bash
# same-box A/B: round-2 tree (worktree _old_r02, its own libcdx.so) against the current tree, whole bench runs
cd /root/repo
for rep in 1 2; do
 for cfg in "cfg2" "cfg5" "cfg2 --dtype fp16"; do
    (cd _old_r02 && timeout -k 10 300 python bench.py --config $cfg --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-sample-call 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg r02', d['value'], d['ms_per_step'])")
    timeout -k 10 300 python bench.py --config $cfg --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-sample-call --no-strict-f32 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg now', d['value'], d['ms_per_step'])"
 done
done
cd /tmp && export TMPDIR=/tmp
cd /root/repo/_old_r02 && rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r3i_old_cfg5 -- python3 bench.py --config cfg5 --steps 10 --warmup 2 --no-cpu-baseline --no-sample-call --no-roofline > /root/repo/gpurun_out/r3i_old_cfg5.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r3i_new_cfg5 -- python3 bench.py --config cfg5 --steps 10 --warmup 2 --no-cpu-baseline --no-sample-call --no-roofline --no-strict-f32 > /root/repo/gpurun_out/r3i_new_cfg5.log 2>&1
echo done
