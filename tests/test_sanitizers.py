"""The product's host path under AddressSanitizer / UndefinedBehaviorSanitizer / LeakSanitizer and ThreadSanitizer
(tools/exp/r05/sanitize_host_path.sh: the whole-program fuzzers, sequential and from 8 threads, the examples and the file
pipeline, each compared with the real reference's bytes).  Builds the library twice (~6 minutes), so it only runs on request:
DJB_RUN_SANITIZERS=1 python -m pytest tests/test_sanitizers.py
Round 5's last session: this run caught a NULL djb_params dereferenced for a lambert and a signed overflow in utia's index."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(os.environ.get("DJB_RUN_SANITIZERS") != "1", reason="slow (two instrumented builds): set DJB_RUN_SANITIZERS=1")
def test_host_path_is_clean_under_the_sanitizers():
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "api_fuzz")):
        pytest.skip("oracle/_ref (the real reference's builds) not available on this machine")
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "exp", "r05", "sanitize_host_path.sh")], capture_output=True, text=True, timeout=3000)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-3000:]
    assert "sanitizer reports: 0 bytes" in out and "ThreadSanitizer reports: 0 bytes" in out, out[-3000:]
    for line in ("api_fuzz 20 seeds + MERL files: reference's bytes", "custom_brdf_fuzz 100 seeds: reference's bytes", "TSan: api_fuzz 32 seeds from 8 threads: reference's bytes"):
        assert line in out, out[-3000:]
