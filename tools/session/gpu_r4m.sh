#!/bin/bash
# round 4, call m: stamped timing ablations of the 16-bit wave-specialised tile with the in-kernel clock (what bounds THAT tile?)
cd "$(dirname "$0")/../.." && . tools/session/r4lib.sh
for n in 10 11 12 13 14 15 10; do step "r4m_stamps16_$n" 200 python tools/ws_stamps.py --fp16 --abl $n; done
