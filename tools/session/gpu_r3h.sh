#!/bin/bash
# same-box A/B of whole bench runs: tuning build with CDX_NO_WS=1 (4-wave kernels) against the wave-specialised routing
cd /root/repo
for rep in 1 2; do
 for cfg in "cfg2" "cfg5" "cfg2 --dtype fp16"; do
  for ws in 1 0; do
    CDX_TUNE=1 CDX_NO_WS=$ws timeout -k 10 300 python bench.py --config $cfg --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-sample-call --no-strict-f32 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg NO_WS=$ws', d['value'], d['ms_per_step'])"
  done
 done
done
