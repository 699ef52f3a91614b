#!/bin/bash
# round 4, call e: phase stamps + in-kernel clock of the 16-bit storage tile (where do ITS MFMA waves wait?)
cd "$(dirname "$0")/../.." && . tools/session/r4lib.sh
step r4e_stamps16 200 python tools/ws_stamps.py --fp16
step r4e_stamps16_256 200 python tools/ws_stamps.py --fp16 --shape 16,256,256,256,0,128
