#!/bin/bash
# long randomised sweeps with fresh seeds on the final build (each case is logged before its launch: FUZZ_TRACE)
cd /root/repo
rm -f gpurun_out/fz*.log
FUZZ_TRACE=gpurun_out/fz32.log timeout -k 10 700 python tools/fuzz_conv.py 1500 201 > gpurun_out/fz32.out 2>&1; echo rc32=$?; tail -1 gpurun_out/fz32.out | cut -c1-300
FUZZ_TRACE=gpurun_out/fz16.log timeout -k 10 400 python tools/fuzz_conv16.py 1000 202 > gpurun_out/fz16.out 2>&1; echo rc16=$?; tail -1 gpurun_out/fz16.out | cut -c1-200
timeout -k 10 300 python tools/fuzz_attn.py 300 204 2>&1 | tail -1
timeout -k 10 500 python tests/fuzz_unet.py 40 203 2>&1 | tail -1
