"""bench.py quotes a PMC traffic digest only for the SAME kernel, the SAME build of the kernel sources and the SAME workload
(VERDICT r01 'What's weak' 4: a figure measured on another build must not pass as this build's).  CPU only."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench


def _digest(d, name, pattern, sha, hbm, config=None):
    body = {"pattern": pattern, "csrc_sha16": sha, "hbm_bytes_per_launch": hbm}
    if config:
        body["config"] = config
    with open(os.path.join(d, name), "w") as f:
        json.dump(body, f)


def test_traffic_digest_selection(tmp_path):
    d = str(tmp_path)
    _digest(d, "r09_a_traffic.json", "KernelA<1>", "aaaa", 100.0)                      # no config field: taken on cfg2
    _digest(d, "r09_b_traffic.json", "KernelA<1>", "bbbb", 200.0, "cfg2")
    _digest(d, "r09_b_traffic_cfg5.json", "KernelB<", "bbbb", 300.0, "cfg5")
    _digest(d, "r09_b_mfma_busy.json", "KernelA<1>", "bbbb", 999.0, "cfg2")              # not a traffic digest
    open(os.path.join(d, "r09_c_traffic.json"), "w").write("{ truncated")                # unreadable: skipped
    name, tr, stale = bench.find_traffic_digest("KernelA<1>", "bbbb", "cfg2", d)
    assert name == "r09_b_traffic.json" and tr["hbm_bytes_per_launch"] == 200.0 and stale is None
    name, tr, stale = bench.find_traffic_digest("KernelA<1>", "aaaa", "cfg2", d)         # the older build's own digest
    assert name == "r09_a_traffic.json" and tr["hbm_bytes_per_launch"] == 100.0
    # another build: nothing is quoted, the newest digest of that kernel and workload is NAMED
    name, tr, stale = bench.find_traffic_digest("KernelA<1>", "cccc", "cfg2", d)
    assert name is None and tr is None and stale == {"file": "profiles/r09_b_traffic.json", "csrc_sha16": "bbbb", "hbm_bytes_per_launch": 200}
    # another workload or another kernel: nothing, not even a stale pointer
    assert bench.find_traffic_digest("KernelA<1>", "bbbb", "cfg4", d) == (None, None, None)
    assert bench.find_traffic_digest("KernelC<", "bbbb", "cfg2", d) == (None, None, None)
    name, tr, _ = bench.find_traffic_digest("KernelB<", "bbbb", "cfg5", d)
    assert name == "r09_b_traffic_cfg5.json" and tr["config"] == "cfg5"

