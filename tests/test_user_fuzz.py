"""Randomised whole-program checks against the REAL reference.
examples/api_fuzz.cpp: the library's own classes through the reference's public interface with random arguments -- both analytic lobes
with the five Fresnel terms and every parameterisation, every operator and query (stray directions included), mutators, the LEAN
representation, sgd / abc, lambert, a UTIA file written by the program, isotropic and anisotropic fits, the error messages.
examples/custom_brdf_fuzz.cpp: the reference's extension points: user-defined BRDFs of five shapes, a
user-defined Fresnel term and a user-defined radial NDF with random coefficients, fitted (tabular, tabular_anisotropic) at random
resolutions and evaluated / sampled at random directions; every printed float must equal the REAL reference's, bit for bit.
  * golden: the first seeds against tests/golden/reftests/{custom_brdf_fuzz,api_fuzz}.txt (written by tests/golden/make_reftests.sh from
    the reference);
  * live: further seeds against oracle/_ref/{custom_brdf_fuzz,api_fuzz} -- the same sources compiled against /root/reference/dj_brdf.h
    by oracle/Makefile (test infrastructure; the binaries travel to the GPU box, the reference's sources do not)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "examples", "custom_brdf_fuzz")
REF = os.path.join(ROOT, "oracle", "_ref", "custom_brdf_fuzz")
GOLDEN = os.path.join(ROOT, "tests", "golden", "reftests", "custom_brdf_fuzz.txt")


def run(exe, first, count, env_extra=None, drop=(), scratch=None, merl=False, threads=0):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {}, DJB_QUIET="1")
    r = subprocess.run([exe, str(first), str(count)] + ([str(scratch)] if scratch else []) + (["merl"] if merl else []) + ([f"threads={threads}"] if threads else []),
                       capture_output=True, timeout=1500, env=env)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    return r.stdout


_REF_OUTPUT = {}


def run_reference(exe, first, count, scratch=None, merl=False):
    """the reference's output for a seed range, run once per session (it does not depend on the scratch directory)"""
    key = (exe, first, count, merl)
    if key not in _REF_OUTPUT:
        _REF_OUTPUT[key] = run(exe, first, count, scratch=scratch, merl=merl)
    return _REF_OUTPUT[key]


def first_difference(a, b):
    la, lb = a.decode().splitlines(), b.decode().splitlines()
    for k, (x, y) in enumerate(zip(la, lb)):
        if x != y:
            ctx = [l for l in la[:k] if l.startswith("== seed")][-1:]
            return f"line {k} ({ctx}): {x!r} != {y!r}"
    return f"{len(la)} vs {len(lb)} lines"


def seed_blocks(text):
    out, cur, key = {}, [], None
    for line in text.decode().splitlines():
        if line.startswith("== seed "):
            if key is not None:
                out[key] = cur
            key, cur = int(line.split()[2]), []
        cur.append(line)
    if key is not None:
        out[key] = cur
    return out


def assert_same_as_reference(got, want, ref_exe, extra=()):
    """got == want, except where the REFERENCE has no defined answer: tabular_anisotropic::qf2 (dj_brdf.h:2812-2822) reads past the end of an
    m_qf2 that compute_qf2 (:3005-3037) left shorter than elev x azim, so what it prints for such a seed depends on the heap, i.e. on the
    seeds run before it (5 of 3200 seeds in profiles/r05/fuzz_soak.txt).  A differing seed is accepted only if the reference, run ALONE,
    disagrees with its own in-sequence output, and we differ from it on those lines only."""
    if got == want:
        return
    G, W = seed_blocks(got), seed_blocks(want)
    assert sorted(G) == sorted(W), first_difference(got, want)
    for k in sorted(W):
        if G[k] == W[k]:
            continue
        alone = seed_blocks(subprocess.run([ref_exe, str(k), "1"] + [str(x) for x in extra], capture_output=True, timeout=600).stdout)[k]
        assert len(alone) == len(W[k]) == len(G[k]), f"seed {k}: {first_difference(got, want)}"
        unstable = {n for n, (a, b) in enumerate(zip(alone, W[k])) if a != b}
        ours_off = {n for n, (a, b) in enumerate(zip(G[k], W[k])) if a != b} | {n for n, (a, b) in enumerate(zip(G[k], alone)) if a != b}
        assert unstable and ours_off <= unstable, f"seed {k}: lines {sorted(ours_off)[:6]} differ from a reproducible reference: {first_difference(got, want)}"


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
    assert_same_as_reference(got, want, REF)


@pytest.mark.gpu
def test_user_fuzz_golden_on_gpu():
    need(EXE)
    got = run(EXE, 1, 6, drop=("DJB_DEVICE",))
    assert got == open(GOLDEN, "rb").read(), first_difference(got, open(GOLDEN, "rb").read())


@pytest.mark.gpu
def test_user_fuzz_live_on_gpu():
    need(EXE); need(REF)
    want, got = run(REF, 2000, 40), run(EXE, 2000, 40, drop=("DJB_DEVICE",))
    assert_same_as_reference(got, want, REF)


# ---------------------------------------------------------------------------------------------- the shipped classes (examples/api_fuzz.cpp)
API_EXE = os.path.join(ROOT, "examples", "api_fuzz")
API_REF = os.path.join(ROOT, "oracle", "_ref", "api_fuzz")
API_GOLDEN = os.path.join(ROOT, "tests", "golden", "reftests", "api_fuzz.txt")


def test_api_fuzz_golden_on_host_path(tmp_path):
    need(API_EXE)
    got = run(API_EXE, 1, 4, {"DJB_DEVICE": "cpu"}, scratch=tmp_path)
    assert got == open(API_GOLDEN, "rb").read(), first_difference(got, open(API_GOLDEN, "rb").read())


def test_api_fuzz_live_on_host_path(tmp_path):
    need(API_EXE); need(API_REF)
    want, got = run(API_REF, 5000, 30, scratch=tmp_path), run(API_EXE, 5000, 30, {"DJB_DEVICE": "cpu"}, scratch=tmp_path)
    assert want.count(b"== seed") == 30
    assert_same_as_reference(got, want, API_REF, (tmp_path,))


@pytest.mark.gpu
def test_api_fuzz_golden_on_gpu(tmp_path):
    need(API_EXE)
    got = run(API_EXE, 1, 4, drop=("DJB_DEVICE",), scratch=tmp_path)
    assert got == open(API_GOLDEN, "rb").read(), first_difference(got, open(API_GOLDEN, "rb").read())


@pytest.mark.gpu
@pytest.mark.parametrize("scalar_on_device", ["0", "1"])
def test_api_fuzz_live_on_gpu(tmp_path, scalar_on_device):
    """one-pair calls answered by the host twin of the GPU objects (default) and sent through the kernels (DJB_SCALAR_ON_DEVICE=1)"""
    need(API_EXE); need(API_REF)
    want = run_reference(API_REF, 6000, 30, scratch=tmp_path)
    got = run(API_EXE, 6000, 30, {"DJB_SCALAR_ON_DEVICE": scalar_on_device}, drop=("DJB_DEVICE",), scratch=tmp_path)
    assert_same_as_reference(got, want, API_REF, (tmp_path,))


def test_api_fuzz_with_merl_files_on_host_path(tmp_path):
    """each seed also writes a 35 MB MERL file (a noisy GGX-like table with invalid bins), looks it up and fits it at a random resolution"""
    need(API_EXE); need(API_REF)
    want, got = run(API_REF, 8000, 4, scratch=tmp_path, merl=True), run(API_EXE, 8000, 4, {"DJB_DEVICE": "cpu"}, scratch=tmp_path, merl=True)
    assert want.count(b"tabular(merl") == 4
    assert_same_as_reference(got, want, API_REF, (tmp_path, "merl"))


@pytest.mark.gpu
@pytest.mark.parametrize("scalar_on_device", ["0", "1"])
def test_api_fuzz_with_merl_files_on_gpu(tmp_path, scalar_on_device):
    need(API_EXE); need(API_REF)
    want = run_reference(API_REF, 9000, 6, scratch=tmp_path, merl=True)
    got = run(API_EXE, 9000, 6, {"DJB_SCALAR_ON_DEVICE": scalar_on_device}, drop=("DJB_DEVICE",), scratch=tmp_path, merl=True)
    assert_same_as_reference(got, want, API_REF, (tmp_path, "merl"))


# ---------------------------------------------------------------------------------------------- concurrency: threads=N
# The reference is a header of const methods: any number of threads may use it.  The facade promises the same on ONE shared
# default context (INTEGRATION.md "Ownership and threading"): the seeds dealt to 8 host threads -- objects created, fitted, queried
# and destroyed concurrently -- must print what the sequential reference prints.
def test_fuzz_programs_from_eight_threads_on_host_path(tmp_path):
    need(API_EXE); need(API_REF); need(EXE); need(REF)
    want = run(API_REF, 12000, 32, scratch=tmp_path)
    got = run(API_EXE, 12000, 32, {"DJB_DEVICE": "cpu"}, scratch=tmp_path, threads=8)
    assert_same_as_reference(got, want, API_REF, (tmp_path,))
    want, got = run(REF, 12000, 32), run(EXE, 12000, 32, {"DJB_DEVICE": "cpu"}, threads=8)
    assert_same_as_reference(got, want, REF)


@pytest.mark.gpu
@pytest.mark.parametrize("scalar_on_device", ["0", "1"])
def test_fuzz_programs_from_eight_threads_on_gpu(tmp_path, scalar_on_device):
    need(API_EXE); need(API_REF); need(EXE); need(REF)
    env = {"DJB_SCALAR_ON_DEVICE": scalar_on_device}
    want = run_reference(API_REF, 13000, 24, scratch=tmp_path, merl=True)
    got = run(API_EXE, 13000, 24, env, drop=("DJB_DEVICE",), scratch=tmp_path, merl=True, threads=8)
    assert_same_as_reference(got, want, API_REF, (tmp_path, "merl"))
    want, got = run_reference(REF, 13000, 40), run(EXE, 13000, 40, env, drop=("DJB_DEVICE",), threads=8)
    assert_same_as_reference(got, want, REF)
