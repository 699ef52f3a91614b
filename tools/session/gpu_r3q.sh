#!/bin/bash
# four-phase upsample convolution: parity (kernel + range + stats tests with upsample cases), then the bench line
cd /root/repo
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_range_gpu.py tests/test_e2e_gpu.py tests/test_context_gpu.py -m gpu -x -q 2>&1 | tail -6
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-sample-call --no-strict-f32 --details > gpurun_out/r3q_bench.json 2> gpurun_out/r3q_bench.err; echo "bench rc=$?"
python -c "import json; d=json.load(open('gpurun_out/r3q_bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['avg_launch_ms'])"
(cd _old_r02 && timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-sample-call 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('r02', d['value'], d['ms_per_step'])")
