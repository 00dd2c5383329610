"""The five Mitsuba BSDF shells (mitsuba/*.cpp) cannot be built here -- the Mitsuba 0.5 SDK is absent -- but they
must at least be well-formed C++ against the djb:: facade (include/djb_hip.hpp) and the BSDF API they use.
`g++ -fsyntax-only` over each shell with tests/mitsuba_mock (a minimal, declaration-only stand-in written from the
shells' own call sites; test infrastructure, proves nothing about Mitsuba itself) on the include path."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHELLS = ["dj_merl", "dj_utia", "dj_abc", "dj_sgd", "dj_beckmannconductor"]


@pytest.mark.parametrize("shell", SHELLS)
def test_shell_is_well_formed(shell):
    cxx = shutil.which("g++") or shutil.which("c++")
    if cxx is None:
        pytest.skip("no host C++ compiler")
    r = subprocess.run([cxx, "-std=c++11", "-fsyntax-only", "-Wall", "-Wno-unused-parameter",
                        "-I", os.path.join(ROOT, "tests", "mitsuba_mock"), "-I", os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "mitsuba", shell + ".cpp")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-4000:]
