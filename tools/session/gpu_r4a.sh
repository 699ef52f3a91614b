#!/bin/bash
# round 4, call a: the scale diagnostics (per-layer error per numerical variant), barrier accounting of the shipped tile, the new
# parity / boundary tests, one full bench line with the parity object, a fuzz sweep with independent additive scales
cd "$(dirname "$0")/../.." && . tools/session/r4lib.sh
step r4a_diag 600 python tools/diag_scale.py
step r4a_stamps 200 python tools/ws_stamps.py
step r4a_stamps_mfma_bound 200 python tools/ws_stamps.py --tile 109
step r4a_tests 1100 python -m pytest tests/test_boundary_gpu.py tests/test_range_gpu.py tests/test_e2e_gpu.py tests/test_bench_contract_gpu.py tests/test_kernels_gpu.py -q -x --timeout 900
step r4a_bench 500 python bench.py --no-parity-gate
step r4a_fuzz 400 python tools/fuzz_conv.py 300 501
