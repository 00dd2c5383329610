"""Randomised check of the reference's extension points (examples/custom_brdf_fuzz.cpp): user-defined BRDFs of five shapes, a
user-defined Fresnel term and a user-defined radial NDF with random coefficients, fitted (tabular, tabular_anisotropic) at random
resolutions and evaluated / sampled at random directions; every printed float must equal the REAL reference's, bit for bit.
  * golden: seeds 1..6 against tests/golden/reftests/custom_brdf_fuzz.txt (written by tests/golden/make_reftests.sh from the reference);
  * live: further seeds against oracle/_ref/custom_brdf_fuzz -- the same source compiled against /root/reference/dj_brdf.h by
    oracle/Makefile (test infrastructure; the binary travels to the GPU box, the reference's sources do not)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "examples", "custom_brdf_fuzz")
REF = os.path.join(ROOT, "oracle", "_ref", "custom_brdf_fuzz")
GOLDEN = os.path.join(ROOT, "tests", "golden", "reftests", "custom_brdf_fuzz.txt")


def run(exe, first, count, env_extra=None, drop=()):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {}, DJB_QUIET="1")
    r = subprocess.run([exe, str(first), str(count)], capture_output=True, timeout=1500, env=env)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    return r.stdout


def first_difference(a, b):
    la, lb = a.decode().splitlines(), b.decode().splitlines()
    for k, (x, y) in enumerate(zip(la, lb)):
        if x != y:
            ctx = [l for l in la[:k] if l.startswith("== seed")][-1:]
            return f"line {k} ({ctx}): {x!r} != {y!r}"
    return f"{len(la)} vs {len(lb)} lines"


def need(path):
    if not os.path.exists(path):
        pytest.skip(f"{os.path.relpath(path, ROOT)} not built")


def test_user_fuzz_golden_on_host_path():
    need(EXE)
    got = run(EXE, 1, 6, {"DJB_DEVICE": "cpu"})
    assert got == open(GOLDEN, "rb").read(), first_difference(got, open(GOLDEN, "rb").read())


def test_user_fuzz_live_on_host_path():
    need(EXE); need(REF)
    want, got = run(REF, 1000, 40), run(EXE, 1000, 40, {"DJB_DEVICE": "cpu"})
    assert want.count(b"== seed") == 40
    assert got == want, first_difference(got, want)


@pytest.mark.gpu
def test_user_fuzz_golden_on_gpu():
    need(EXE)
    got = run(EXE, 1, 6, drop=("DJB_DEVICE",))
    assert got == open(GOLDEN, "rb").read(), first_difference(got, open(GOLDEN, "rb").read())


@pytest.mark.gpu
def test_user_fuzz_live_on_gpu():
    need(EXE); need(REF)
    want, got = run(REF, 2000, 40), run(EXE, 2000, 40, drop=("DJB_DEVICE",))
    assert got == want, first_difference(got, want)
