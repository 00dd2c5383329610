"""The facade as a drop-in for the reference's OWN callers: the six Mitsuba plugin sources of the reference
(/root/reference/mitsuba/{dj_merl,dj_utia,dj_abc,dj_sgd,dj_beckmannconductor,dj_brdf}.cpp), compiled UNCHANGED against
include/dj_brdf.h + libdjb_hip.so and the functional stand-in of the Mitsuba API (tests/mitsuba_mock), must behave exactly as they
do on the reference's own header: every record of tests/golden/shells.npz (the five named plugins; GLSL programs included -- it is
the reference's shader code that is compiled here) and of tests/golden/shells_dj_brdf.npz (the sixth, dj_brdf.cpp: distribution
beckmann / ggx / tabular, lobes fitted from MERL and UTIA files at the plugin's 90 / 90 x 90 resolutions, both Fresnel modes,
constructor errors).  The libraries are built in the build container (oracle/_ref/shells_on_facade/, by build()) and travel to the
GPU box; the reference's sources do not.  Like tests/test_mitsuba_shells.py this pins glue and interface, relative to the mock."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = os.path.join(ROOT, "tests", "mitsuba_mock", "shell_cases.py")
LIBS = os.path.join(ROOT, "oracle", "_ref", "shells_on_facade")
SIX = ["dj_merl", "dj_utia", "dj_abc", "dj_sgd", "dj_beckmannconductor", "dj_brdf"]


def run_side(tmp_path, which, env_extra, drop=()):
    if not all(os.path.exists(os.path.join(LIBS, f"libshell_{s}.so")) for s in SIX):
        pytest.skip("oracle/_ref/shells_on_facade not built (needs /root/reference at build time)")
    out = str(tmp_path / f"refsrc_{which}.npz")
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra, DJB_QUIET="1")
    r = subprocess.run([sys.executable, CASES, "--side", "refsrc", "--which", which, "--prebuilt", "--out", out], capture_output=True, text=True, env=env, timeout=1500)
    assert r.returncode == 0, (r.stdout + r.stderr)[-6000:]
    return np.load(out)


def compare(got, want):
    assert sorted(got.files) == sorted(want.files), sorted(set(got.files) ^ set(want.files))[:20]
    bad = []
    for k in want.files:
        g, w = got[k], want[k]
        if w.dtype.kind in "US":
            if str(g) != str(w):
                bad.append(f"{k}: {str(g)[:120]!r} != {str(w)[:120]!r}")
        elif w.dtype.kind == "f":
            if g.shape != w.shape or not np.array_equal(g.view(np.uint32), w.view(np.uint32)):
                bad.append(f"{k}: values differ")
        elif g.shape != w.shape or not np.array_equal(g, w):
            bad.append(f"{k}: {g} != {w}")
    assert not bad, f"{len(bad)} differences from the reference's behaviour:\n" + "\n".join(bad[:40])


GOLDEN = {"five": os.path.join(ROOT, "tests", "golden", "shells.npz"), "dj_brdf": os.path.join(ROOT, "tests", "golden", "shells_dj_brdf.npz")}


def test_sixth_plugin_golden_is_substantial():
    g = np.load(GOLDEN["dj_brdf"])
    names = sorted(set(k.split("/")[0] for k in g.files if "/" in k))
    assert len(names) >= 30
    live = [n for n in names if f"{n}/eval" in g.files]
    assert len(live) >= 22 and all(np.count_nonzero(g[f"{n}/eval"]) for n in live if "err" not in n)
    assert not np.array_equal(g["db_ggx_merl/eval"], g["db_beckmann_merl/eval"]) and not np.array_equal(g["db_tabular_merl/eval"], g["db_ggx_merl/eval"])
    assert not np.array_equal(g["db_tabular_utia/eval"], g["db_tabular_merl/eval"])
    assert "tabular distribution requires a merl file" in str(g["db_tabular_nofile/str_create_error"])


@pytest.mark.parametrize("which", ["five", "dj_brdf"])
def test_reference_plugin_sources_on_the_facade_host_path(tmp_path, which):
    compare(run_side(tmp_path, which, {"DJB_DEVICE": "cpu"}), np.load(GOLDEN[which]))


@pytest.mark.gpu
@pytest.mark.parametrize("scalar_on_device", ["0", "1"])
@pytest.mark.parametrize("which", ["five", "dj_brdf"])
def test_reference_plugin_sources_on_the_facade_gpu(tmp_path, which, scalar_on_device):
    compare(run_side(tmp_path, which, {"DJB_SCALAR_ON_DEVICE": scalar_on_device}, drop=("DJB_DEVICE",)), np.load(GOLDEN[which]))


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="build container only")
def test_sixth_plugin_golden_is_what_the_reference_produces(tmp_path):
    out = str(tmp_path / "ref_db.npz")
    r = subprocess.run([sys.executable, CASES, "--side", "ref", "--which", "dj_brdf", "--out", out], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    compare(np.load(out), np.load(GOLDEN["dj_brdf"]))
