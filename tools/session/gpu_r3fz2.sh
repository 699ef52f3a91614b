#!/bin/bash
# longer randomised sweeps on the final build incl. the conv_out mode (tile 12), fresh seeds
cd /root/repo
rm -f gpurun_out/fz*.log
FUZZ_TRACE=gpurun_out/fzco.log timeout -k 10 500 python tools/fuzz_conv.py 500 301 convout > gpurun_out/fzco.out 2>&1; echo rcco=$?; tail -1 gpurun_out/fzco.out | cut -c1-300
FUZZ_TRACE=gpurun_out/fz32.log timeout -k 10 500 python tools/fuzz_conv.py 800 302 > gpurun_out/fz32.out 2>&1; echo rc32=$?; tail -1 gpurun_out/fz32.out | cut -c1-300
