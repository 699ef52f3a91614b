#!/bin/bash
# scratch build of the tuning library: s_setprio 3 for the MFMA waves (CDX_EXP=1) or for the producer waves (CDX_EXP=2) of the
# wave-specialised split tiles; whole steps on one box, interleaved
cd /root/repo
export CDX_TUNE=1
B="python bench.py --steps 40 --no-cpu-baseline --no-sample-call --no-strict-f32 --no-roofline"
for e in 0 1 2 0 1 2 0; do
  CDX_EXP=$e $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('EXP $e cfg2', d['value'], d['ms_per_step'])"
done
for e in 0 1 2; do
  echo "EXP $e"; CDX_EXP=$e timeout -k 10 200 python tools/conv_bench.py --shape 16,256,256,128,0,128,3,1 --tiles 11 --gn --stats --rounds 3 2>&1 | tail -1
done
