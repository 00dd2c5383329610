"""djb:: objects with static storage duration (globals of a plugin or a small tool, function-local statics): constructed before
main() -- the process's default context comes to life inside the first constructor -- and destroyed at exit, after main() has
returned, in an order the program does not control.  The program must print the reference's values and exit cleanly (no
crash in the HIP runtime's own teardown)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "dj_brdf_amd", "lib")
PROG = r"""
#include <cstdio>
#define DJ_BRDF_IMPLEMENTATION 1
#include "dj_brdf.h"
static djb::ggx g_ggx(djb::fresnel::schlick(djb::vec3(0.9f, 0.5f, 0.2f)));
static djb::beckmann g_bk;
static djb::tabular g_tab(g_bk, 24);
static djb::sgd g_sgd("chrome");
int main()
{
	djb::vec3 i(0.3f, 0.2f, 0.9f), o(-0.2f, 0.4f, 0.8f);
	printf("%a %a %a\n", g_ggx.eval(i, o).x, g_tab.pdf(i, o), g_sgd.eval(i, o).y);
	static djb::lambert late;
	printf("%a\n", late.eval(i, o).x);
	return 0;
}
"""
WANT = "0x1.6e94dep-4 0x1.73920ap-4 0x1.116342p-9\n0x1.45f306p-2\n"      # what the same program prints on the reference's header


def build(tmp_path):
    f = tmp_path / "glob.cpp"; f.write_text(PROG)
    exe = tmp_path / "glob"
    r = subprocess.run(["g++", "-O1", "-std=c++11", "-DNVERBOSE", "-I" + os.path.join(ROOT, "include"), "-o", str(exe), str(f), "-L" + LIBDIR, "-ldjb_hip",
                        "-Wl,-rpath," + LIBDIR, "-Wl,-rpath-link,/opt/rocm/lib", "-pthread"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_static_objects_on_host_path(tmp_path):
    out = subprocess.run([str(build(tmp_path))], env=dict(os.environ, DJB_DEVICE="cpu", DJB_QUIET="1"), capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout == WANT, (out.returncode, out.stdout, out.stderr[-500:])


@pytest.mark.gpu
@pytest.mark.parametrize("scalar_on_device", ["0", "1"])
def test_static_objects_on_gpu(tmp_path, scalar_on_device):
    env = {k: v for k, v in os.environ.items() if k != "DJB_DEVICE"}
    env.update(DJB_QUIET="1", DJB_SCALAR_ON_DEVICE=scalar_on_device)
    out = subprocess.run([str(build(tmp_path))], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout == WANT, (out.returncode, out.stdout, out.stderr[-500:])
