#!/bin/bash
# 1-D XCD-aware grid for layers with several 128-channel blocks: parity, bench cfg2 / cfg4, PMC traffic of cfg4
cd /root/repo
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_range_gpu.py tests/test_e2e_gpu.py tests/test_configs_gpu.py tests/test_fp16_gpu.py -m gpu -x -q 2>&1 | tail -3
python bench.py --steps 30 --no-cpu-baseline --no-sample-call --no-strict-f32 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg2', d['value'], d['ms_per_step'])"
python bench.py --config cfg4 --steps 10 --warmup 2 --no-cpu-baseline --no-sample-call --no-strict-f32 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg4', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
ARGS="--config cfg4 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-sample-call --no-strict-f32"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /root/repo/gpurun_out/r3w4_fetch -- python3 /root/repo/bench.py $ARGS > /root/repo/gpurun_out/r3w4_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /root/repo/gpurun_out/r3w4_write -- python3 /root/repo/bench.py $ARGS > /root/repo/gpurun_out/r3w4_write.log 2>&1
echo rc=$?
