#!/bin/bash
# wave-specialised product build: parity suites of every dtype + bench lines cfg2 / cfg5 / cfg4 / cfg2 fp16
cd /root/repo
mkdir -p gpurun_out
timeout -k 10 1500 python -m pytest tests/test_kernels_gpu.py tests/test_range_gpu.py tests/test_e2e_gpu.py tests/test_fp16_gpu.py tests/test_bf16_gpu.py tests/test_configs_gpu.py -m gpu -x -q > gpurun_out/r3f_suite.log 2>&1; echo "suite rc=$?"
tail -4 gpurun_out/r3f_suite.log
timeout -k 10 600 python bench.py --steps 30 --warmup 3 --no-cpu-baseline > gpurun_out/r3f_bench_cfg2.json 2> gpurun_out/r3f_bench_cfg2.err; echo "cfg2 rc=$?"; head -c 400 gpurun_out/r3f_bench_cfg2.json; echo
timeout -k 10 600 python bench.py --config cfg5 --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r3f_bench_cfg5.json 2> gpurun_out/r3f_bench_cfg5.err; echo "cfg5 rc=$?"; head -c 400 gpurun_out/r3f_bench_cfg5.json; echo
timeout -k 10 600 python bench.py --config cfg2 --dtype fp16 --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r3f_bench_cfg2_fp16.json 2>/dev/null; echo "cfg2fp16 rc=$?"; head -c 400 gpurun_out/r3f_bench_cfg2_fp16.json; echo
timeout -k 10 600 python bench.py --config cfg4 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r3f_bench_cfg4.json 2>/dev/null; echo "cfg4 rc=$?"; head -c 400 gpurun_out/r3f_bench_cfg4.json; echo
