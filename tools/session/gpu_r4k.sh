#!/bin/bash
# round 4, call k: roofline.traffic measured live by bench.py itself (two rocprofv3 --pmc child runs): the new contract test, then the default command, timed
cd "$(dirname "$0")/../.." && . tools/session/r4lib.sh
step r4k_test 600 python -m pytest tests/test_bench_contract_gpu.py -q -x --timeout 500
SECONDS=0
step r4k_bench 700 python bench.py
echo "default bench.py wall: $SECONDS s"
