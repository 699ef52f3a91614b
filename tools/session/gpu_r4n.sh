#!/bin/bash
# round 4, call n: what is the producers' path to the first barrier made of?  (launch + setup, halo round trip, staging, barrier)
cd "$(dirname "$0")/../.." && . tools/session/r4lib.sh
step r4n_split 200 python tools/ws_stamps.py
step r4n_fp16 200 python tools/ws_stamps.py --fp16
