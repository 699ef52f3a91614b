#!/bin/bash
# randomised sweeps on the FINAL source hash (wave-specialised 64-pixel tile, interleaved conv_out loop), fresh seeds
cd /root/repo
rm -f gpurun_out/fz*.log
FUZZ_TRACE=gpurun_out/fz32.log timeout -k 10 500 python tools/fuzz_conv.py 900 401 > gpurun_out/fz32.out 2>&1; echo rc32=$?; tail -1 gpurun_out/fz32.out | cut -c1-300
FUZZ_TRACE=gpurun_out/fzco.log timeout -k 10 300 python tools/fuzz_conv.py 300 402 convout > gpurun_out/fzco.out 2>&1; echo rcco=$?; tail -1 gpurun_out/fzco.out | cut -c1-300
timeout -k 10 400 python tests/fuzz_unet.py 30 403 2>&1 | tail -1
