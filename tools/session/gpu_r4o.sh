#!/bin/bash
# round 4, call o: producers' prologue at raised wave priority -- in-process A/B (tile 124 = without, 11 = with; 16-bit: abl 11 = without), stamps
cd "$(dirname "$0")/../.." && . tools/session/r4lib.sh
for shape in 16,256,256,128,0,128,3,1 16,256,256,128,128,128,3,1 16,128,128,256,0,256,3,1 16,64,64,256,0,256,3,1 8,512,512,192,0,192,3,1; do
    step "r4o_ab_${shape//,/_}" 300 python tools/conv_bench.py --shape $shape --tiles 124,11,124,11 --gn --stats --check --rounds 5
done
step r4o_ab16_a 200 python tools/conv16_bench.py --shape 16,256,256,128,0,128,3,1 --abl 11,0,11,0
step r4o_ab16_b 200 python tools/conv16_bench.py --shape 16,256,256,256,0,128,3,1 --abl 11,0,11,0
step r4o_ab16_c 200 python tools/conv16_bench.py --shape 16,128,128,256,0,256,3,1 --abl 11,0,11,0
step r4o_stamps 200 python tools/ws_stamps.py
step r4o_stamps16 200 python tools/ws_stamps.py --fp16
