#!/bin/bash
# packed-f32 staging in the producer waves: isolated A/B against the 4-wave tile, then whole bench cfg2 / cfg5 next to the round-2 tree
cd /root/repo
python tools/conv_bench.py --shape 16,256,256,128,0,128,3,1 --tiles 11,106 --gn --stats --check --rounds 7 --iters 10 2>&1 | grep -v amdgpu.ids
python tools/conv16_bench.py --abl 0 --shape 16,256,256,128,0,128,3,1 2>&1 | grep -v amdgpu.ids
for cfg in "cfg2" "cfg5"; do
    timeout -k 10 300 python bench.py --config $cfg --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-sample-call --no-strict-f32 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg now', d['value'], d['ms_per_step'])"
    (cd _old_r02 && timeout -k 10 300 python bench.py --config $cfg --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-sample-call 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg r02', d['value'], d['ms_per_step'])")
done
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_fp16_gpu.py tests/test_bf16_gpu.py -m gpu -x -q 2>&1 | tail -3
