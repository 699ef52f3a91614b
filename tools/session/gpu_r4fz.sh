#!/bin/bash
# round 4: long randomised sweeps on the final source hash, fresh seeds (every case logged before its launch: FUZZ_TRACE)
cd "$(dirname "$0")/../.." && . tools/session/r4lib.sh
export FUZZ_TRACE=gpurun_out/r4fz_trace.log
step r4fz_conv 900 python tools/fuzz_conv.py 1500 601
step r4fz_convout 400 python tools/fuzz_conv.py 400 602 convout
step r4fz_conv16 700 python tools/fuzz_conv16.py 1000 603
step r4fz_attn 400 python tools/fuzz_attn.py 300 604
step r4fz_unet 900 python tests/fuzz_unet.py 30 605
rm -f gpurun_out/r4fz_trace.log
