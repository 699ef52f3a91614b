#!/bin/bash
# round 4, call g: is the MFMA waves' wait for the first chunk the producers' path or their own residual fetch?  stamps with / without residual
cd "$(dirname "$0")/../.." && . tools/session/r4lib.sh
step r4g_split_res 200 python tools/ws_stamps.py
step r4g_split_nores 200 python tools/ws_stamps.py --nores
step r4g_fp16_res 200 python tools/ws_stamps.py --fp16
step r4g_fp16_nores 200 python tools/ws_stamps.py --fp16 --nores
