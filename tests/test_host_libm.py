"""The host-libm hazard: the kernels reproduce glibc 2.35's libm, the host path
calls the HOST's libm.  On a host with another libm the product notices (djb_ctx_libm_matches_host flips, one line on
stderr) and its host path switches to the kernels' restatements compiled for the host, so its values stay the
reference-on-glibc-2.35 values that the GPU batches return.  Simulated here by LD_PRELOADing a libm whose expf / atan2 /
cos are one unit in the last place off."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SHIM = r"""
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdint.h>
#include <string.h>
static float flipf(float y) { uint32_t u; memcpy(&u, &y, 4); if (y == y && (u & 0x7f800000u) != 0x7f800000u && (u & 0x7fffffffu)) u ^= 1u; memcpy(&y, &u, 4); return y; }
static double flipd(double y) { uint64_t u; memcpy(&u, &y, 8); if (y == y && (u >> 52 & 0x7ff) != 0x7ff && (u << 1)) u ^= 1ull; memcpy(&y, &u, 8); return y; }
float expf(float x) { static float (*f)(float); if (!f) f = (float (*)(float))dlsym(RTLD_NEXT, "expf"); return flipf(f(x)); }
double atan2(double y, double x) { static double (*f)(double, double); if (!f) f = (double (*)(double, double))dlsym(RTLD_NEXT, "atan2"); return flipd(f(y, x)); }
double cos(double x) { static double (*f)(double); if (!f) f = (double (*)(double))dlsym(RTLD_NEXT, "cos"); return flipd(f(x)); }
"""


@pytest.fixture(scope="module")
def shim(tmp_path_factory):
    d = tmp_path_factory.mktemp("libmshim")
    src, so = d / "shim.c", d / "libperturbed_m.so"
    src.write_text(SHIM)
    subprocess.run(["gcc", "-O1", "-shared", "-fPIC", "-o", str(so), str(src), "-ldl"], check=True)
    return str(so)


def replay(mode, env_extra):
    env = dict(os.environ); env.pop("DJB_HOST_LIBM", None); env.update(env_extra)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "libm_replay.py"), mode], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1]), r.stderr


def test_this_host_matches_and_uses_its_own_libm():
    out, err = replay("cpu", {})
    assert out["status"] == [1, 0, 1], out          # matches, host libm in use, atan/log known answers fine
    assert out["microfacet_values_differing_from_reference_goldens"] == 0 and out["merl_indices_differing"] == 0
    assert "differs from glibc" not in err


def test_restated_mode_reproduces_the_reference_goldens():
    out, _ = replay("cpu", {"DJB_HOST_LIBM": "restated"})
    assert out["status"][1] == 1
    assert out["microfacet_values_differing_from_reference_goldens"] == 0 and out["merl_indices_differing"] == 0


def test_perturbed_host_libm_is_noticed_and_replaced(shim):
    out, err = replay("cpu", {"LD_PRELOAD": shim})
    assert out["status"][0] == 0 and out["status"][1] == 1, out      # (i) the flag flips, the restatements take over
    assert "differs from glibc 2.35" in err and "expf" in err and "atan2" in err and "cos" in err
    # (ii) the host path still returns the reference's (glibc 2.35) values: what the GPU batches return
    assert out["microfacet_values_differing_from_reference_goldens"] == 0 and out["merl_indices_differing"] == 0


def test_the_perturbation_has_teeth(shim):
    """control: forced back onto the perturbed host libm, the same replay does NOT reproduce the goldens"""
    out, err = replay("cpu", {"LD_PRELOAD": shim, "DJB_HOST_LIBM": "host"})
    assert out["status"][0] == 0 and out["status"][1] == 0
    assert out["microfacet_values_differing_from_reference_goldens"] > 0


@pytest.mark.gpu
def test_scalar_path_equals_batch_path_under_a_perturbed_host_libm(shim):
    out, err = replay("gpu", {"LD_PRELOAD": shim})
    assert out["status"][0] == 0 and out["status"][1] == 1
    assert out["scalar_vs_batch_values_differing"] == 0, out
    base, _ = replay("gpu", {})
    assert base["status"][:2] == [1, 0] and base["scalar_vs_batch_values_differing"] == 0
