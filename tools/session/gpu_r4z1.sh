#!/bin/bash
# round 4: the whole GPU suite on the tree as it stands
cd "$(dirname "$0")/../.." && . tools/session/r4lib.sh
rm -f gpurun_out/metrics.jsonl
step r4z1_suite 1150 python -m pytest tests -m gpu -q --durations=12 --timeout 900
