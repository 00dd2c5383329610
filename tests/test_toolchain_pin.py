"""The toolchain pin (tools/toolchain_pin.py): thirteen trig sites are bit-identical "by exhaustion ... valid for this ROCm release +
glibc 2.35" (INTEGRATION.md section 5).  A toolchain bump must FAIL here until the sweeps are re-run, not silently invalidate them."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import toolchain_pin as T


def test_this_machine_is_the_pinned_toolchain():
    pin = json.load(open(T.PIN))
    d = T.differences(T.current(), pin)
    assert not d, "the toolchain differs from the one the exhaustive trig sweeps were produced on:\n  " + "\n  ".join(d) + "\n" + T.HOWTO


def test_the_library_was_built_with_the_pinned_toolchain():
    assert os.path.exists(T.BUILT), "dj_brdf_amd/lib/toolchain.json missing: run __graft_entry__.build()"
    built, pin = json.load(open(T.BUILT)), json.load(open(T.PIN))
    d = T.differences(built, pin)
    assert not d, "libdjb_hip.so was built with another toolchain than the pinned one:\n  " + "\n  ".join(d) + "\n" + T.HOWTO
