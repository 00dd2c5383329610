"""The decided fast tier of the sgd / abc chains (csrc/djb_fast_models.inc) compiled FOR THE HOST (tools/sgd_fast_check.cpp: the same
source in its host-restated instantiation) against the host's glibc -- the reference's own pow / exp / acos -- and __float128:
no decided value may differ from the reference's float, and the measured distance must stay below the bound (the tool's exit code)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_fast_tier_against_host_glibc(tmp_path):
    exe = str(tmp_path / "sgd_fast_check")
    cc = subprocess.run(["g++", "-O2", "-std=c++17", "-mfma", "-ffp-contract=off", "-fopenmp", "-I", os.path.join(ROOT, "dj_brdf_amd", "csrc"),
                         os.path.join(ROOT, "tools", "sgd_fast_check.cpp"), "-o", exe, "-lquadmath"], capture_output=True, text=True)
    if cc.returncode != 0 and "quadmath" in cc.stderr:
        pytest.skip("no libquadmath on this host")
    assert cc.returncode == 0, cc.stderr[-2000:]
    r = subprocess.run([exe, os.path.join(ROOT, "dj_brdf_amd", "data", "sgd_params.csv"), "20000",
                        os.path.join(ROOT, "dj_brdf_amd", "data", "abc_params.csv"), "4000000"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "MISMATCH" not in r.stdout
    for term in ("g1 :", "ndf:", "abc:"):
        line = [l for l in r.stdout.splitlines() if l.startswith(term)][0]
        assert " 0 decided-but-different" in line, line
