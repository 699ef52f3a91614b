#!/bin/bash
# round 4, call i: what does the CLOCK do when a part of the tile's work is taken away?  stamped timing ablations of the shipped float32 tile
# (108 full, 109 producers stage chunk 0 only, 119 no LDS operand reads, 120 no weight refills, 121 all three, 122 no epilogue)
cd "$(dirname "$0")/../.." && . tools/session/r4lib.sh
for t in 108 109 119 120 121 122 108; do step "r4i_stamps_$t" 200 python tools/ws_stamps.py --tile $t; done
