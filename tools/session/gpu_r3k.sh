#!/bin/bash
cd /root/repo
python tools/conv16_bench.py --plain --abl 0 --shape 16,256,256,128,0,128,3,1 2>&1 | grep -v amdgpu.ids
for rep in 1 2; do
 for cfg in "cfg2" "cfg5"; do
    (cd _old_r02 && timeout -k 10 300 python bench.py --config $cfg --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-sample-call 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg r02', d['value'], d['ms_per_step'])")
    timeout -k 10 300 python bench.py --config $cfg --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-sample-call --no-strict-f32 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg now', d['value'], d['ms_per_step'])"
 done
done
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_range_gpu.py tests/test_fp16_gpu.py -m gpu -x -q 2>&1 | tail -3
