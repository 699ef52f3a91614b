"""Randomised parity sweeps (tools/fuzz_conv.py, tools/fuzz_conv16.py, tools/fuzz_attn.py) as part of the GPU suite: random layer shapes and
fusion flags through the library's own tile choice, against float64 torch on the CPU.  The fp32 sweep found the
`cdx_conv_stats_slots` inconsistency for cout <= 4 in round 1."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("tool,cases,seed,mode", [("fuzz_conv.py", 150, 11, ""), ("fuzz_conv.py", 60, 15, "convout"), ("fuzz_conv16.py", 100, 12, ""),
                                                  ("fuzz_attn.py", 80, 13, ""), ("fuzz_unet.py", 8, 14, "")])
def test_randomised_conv_sweep(lib, tool, cases, seed, mode):
    path = os.path.join(ROOT, "tests" if tool == "fuzz_unet.py" else "tools", tool)
    r = subprocess.run([sys.executable, path, str(cases), str(seed)] + ([mode] if mode else []),
                       capture_output=True, text=True, timeout=900)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert f" 0 bad of {cases}" in r.stdout, tail
