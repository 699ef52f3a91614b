#!/bin/bash
# experiments in a scratch build of the tuning library (not in the tracked sources): CDX_EXP bit 1 = wave-specialised 64-pixel split
# tile (16^2 levels), 2 = gn_finalize2 with 4 loads in flight, 4 = conv_out with its vector work between the MFMA groups
cd /root/repo
export CDX_TUNE=1
CDX_EXP=7 timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_range_gpu.py tests/test_e2e_gpu.py -m gpu -q -x > gpurun_out/r3exp_suite.log 2>&1; echo "suite rc=$?"; tail -3 gpurun_out/r3exp_suite.log
B="python bench.py --steps 40 --no-cpu-baseline --no-sample-call --no-strict-f32 --no-roofline"
for e in 0 7 0 7 1 2 4 0; do
  CDX_EXP=$e $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('EXP $e cfg2', d['value'], d['ms_per_step'])"
done
for e in 0 7 0 7; do
  CDX_EXP=$e $B --config cfg5 --steps 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('EXP $e cfg5', d['value'], d['ms_per_step'])"
done
