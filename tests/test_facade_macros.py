"""The reference header's configuration macros (dj_brdf.h:7-12, 43-48, 552-560) in the facade (include/djb_hip.hpp):
DJB_USE_DOUBLE_PRECISION is refused at compile time, a user DJB_ASSERT fires at the reference's argument checks, a user
DJB_LOG receives the constructors' progress lines unless NVERBOSE is defined.  CPU only (DJB_DEVICE=cpu)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "dj_brdf_amd", "lib")

PROG = r"""
#include <stdexcept>
#include <string>
#include <vector>
#include <cstdarg>
#include <cstdio>
static std::vector<std::string> g_log;
static void my_log(const char *fmt, ...) { char b[256]; va_list a; va_start(a, fmt); vsnprintf(b, sizeof b, fmt, a); va_end(a); g_log.push_back(b); }
#define DJB_ASSERT(x) do { if (!(x)) throw std::runtime_error(std::string("user assert: ") + #x); } while (0)
#define DJB_LOG(format, ...) my_log(format, ##__VA_ARGS__)
#include "dj_brdf.h"
int main()
{
	int fired = 0;
	try { djb::microfacet::params::elliptic(-1.0f, 0.3f); } catch (const std::runtime_error &e) { ++fired; printf("%s\n", e.what()); }
	try { djb::microfacet::params::pdfparams(0.3f, 0.3f, 1.5f); } catch (const std::runtime_error &e) { ++fired; printf("%s\n", e.what()); }
	try { djb::ggx g; djb::tabular t(g, 2); } catch (const std::runtime_error &e) { ++fired; printf("%s\n", e.what()); }
	djb::ggx g;
	djb::tabular t(g, 16);
	for (size_t k = 0; k < g_log.size(); ++k) printf("LOG %s", g_log[k].c_str());
	printf("fired=%d logs=%d\n", fired, (int)g_log.size());
	return 0;
}
"""


def build(tmp_path, src, flags=()):
    f = tmp_path / "prog.cpp"; f.write_text(src)
    exe = tmp_path / "prog"
    r = subprocess.run(["g++", "-O1", "-std=c++14", "-I" + os.path.join(ROOT, "include"), *flags, "-o", str(exe), str(f), "-L" + LIBDIR, "-ldjb_hip",
                        "-Wl,-rpath," + LIBDIR, "-Wl,-rpath-link,/opt/rocm/lib", "-pthread"], capture_output=True, text=True)
    return r, exe


def test_double_precision_build_is_refused(tmp_path):
    r, _ = build(tmp_path, '#include "dj_brdf.h"\nint main() { return 0; }\n', ["-DDJB_USE_DOUBLE_PRECISION=1"])
    assert r.returncode != 0 and "DJB_USE_DOUBLE_PRECISION=1 is not supported" in r.stderr
    r, _ = build(tmp_path, '#include "dj_brdf.h"\nint main() { return sizeof(djb::float_t) == 4 ? 0 : 1; }\n', ["-DDJB_USE_DOUBLE_PRECISION=0"])
    assert r.returncode == 0, r.stderr


def test_user_assert_and_log_macros(tmp_path):
    r, exe = build(tmp_path, PROG)
    assert r.returncode == 0, r.stderr
    env = dict(os.environ, DJB_DEVICE="cpu", DJB_QUIET="1")
    out = subprocess.run([str(exe)], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert "user assert: a1 > 0.0 && a2 > 0.0" in out.stdout and "Invalid correlation parameter" in out.stdout and "Invalid Resolution" in out.stdout
    assert "fired=3 logs=4" in out.stdout
    for line in ("Projected area term ready", "Fresnel function ready", "Slope CDF ready", "Slope QF ready"):
        assert "LOG djb_verbose: " + line in out.stdout
    # NVERBOSE silences the progress lines, as in the reference
    r, exe = build(tmp_path, PROG, ["-DNVERBOSE"])
    out = subprocess.run([str(exe)], env=env, capture_output=True, text=True, timeout=120)
    assert "fired=3 logs=0" in out.stdout


def test_facade_has_the_reference_public_surface():
    """tests/api/api_surface_probe.cpp uses every public name of dj_brdf.h:41-537 once with the reference's signatures -- the classes a
    user may DERIVE from (brdf, fresnel::impl, radial, microfacet) included.  It must pass -fsyntax-only against include/dj_brdf.h; where
    the reference is mounted it is compiled against the reference header as well, which proves the probe itself."""
    src = os.path.join(ROOT, "tests", "api", "api_surface_probe.cpp")
    r = subprocess.run(["g++", "-fsyntax-only", "-DNVERBOSE", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), src], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]
    if os.path.exists("/root/reference/dj_brdf.h"):
        r = subprocess.run(["g++", "-fsyntax-only", "-DNVERBOSE", "-w", "-I", "/root/reference", src], capture_output=True, text=True)
        assert r.returncode == 0, "the probe does not compile against the reference itself:\n" + r.stderr[-4000:]


def test_epsilon_macro(tmp_path):
    """DJB_EPSILON (dj_brdf.h:49-51): defined by the header with the reference's default; a program that redefines it is refused when
    it creates its first object (the kernels implement h.z > 1e-4), not silently given the default's results"""
    prog = '#include <cstdio>\n#include "dj_brdf.h"\nint main() { try { djb::ggx g; printf("eps %g pi %.3f\\n", (double)(djb::float_t)DJB_EPSILON * 1.0, M_PI); }' \
           ' catch (const djb::exc &e) { printf("%s\\n", e.what()); return 3; } return 0; }\n'
    prog = prog.replace("(djb::float_t)DJB_EPSILON", "[]{ using djb::float_t; return DJB_EPSILON; }()")
    env = dict(os.environ, DJB_DEVICE="cpu", DJB_QUIET="1")
    r, exe = build(tmp_path, prog)
    assert r.returncode == 0, r.stderr
    out = subprocess.run([str(exe)], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "eps 0.0001 pi 3.142" in out.stdout, out.stdout + out.stderr
    r, exe = build(tmp_path, prog, ["-DDJB_EPSILON=(float_t)1e-3"])
    assert r.returncode == 0, r.stderr
    out = subprocess.run([str(exe)], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 3 and "DJB_EPSILON was redefined" in out.stdout, out.stdout + out.stderr


def test_header_in_several_translation_units(tmp_path):
    """the reference is included everywhere and implemented in ONE translation unit (`#define DJ_BRDF_IMPLEMENTATION 1`, dj_brdf.h:4-6);
    the facade is header-only: the same two files must compile, link without duplicate symbols, and run"""
    (tmp_path / "a.cpp").write_text('#define DJ_BRDF_IMPLEMENTATION 1\n#include "dj_brdf.h"\nfloat other(const djb::vec3 &i, const djb::vec3 &o);\n'
                                    'int main() { djb::ggx g; djb::vec3 i(0.3f, 0.1f, 0.9f), o(0.1f, 0.2f, 0.95f); printf("%a %a\\n", g.eval(i, o).x, other(i, o)); return 0; }\n')
    (tmp_path / "b.cpp").write_text('#include "dj_brdf.h"\nfloat other(const djb::vec3 &i, const djb::vec3 &o) { djb::beckmann b; djb::tabular t(b, 16); return t.pdf(i, o); }\n')
    for f in ("a", "b"):
        r = subprocess.run(["g++", "-O1", "-std=c++14", "-DNVERBOSE", "-I" + os.path.join(ROOT, "include"), "-c", str(tmp_path / (f + ".cpp")), "-o", str(tmp_path / (f + ".o"))],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
    r = subprocess.run(["g++", str(tmp_path / "a.o"), str(tmp_path / "b.o"), "-o", str(tmp_path / "prog"), "-L" + LIBDIR, "-ldjb_hip", "-Wl,-rpath," + LIBDIR,
                        "-Wl,-rpath-link,/opt/rocm/lib", "-pthread"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = subprocess.run([str(tmp_path / "prog")], env=dict(os.environ, DJB_DEVICE="cpu", DJB_QUIET="1"), capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.strip() == "0x1.6d4a2ap-4 0x1.5af378p-4", out.stdout + out.stderr      # what the reference prints for it


def test_overriding_the_ndf_of_a_resident_lobe_is_a_compile_error(tmp_path):
    """In the reference a class derived from djb::ggx that overrides p22_radial changes eval / pdf / sample (virtual all the way down).
    The library's lobes are answered from HBM and would ignore the override: the facade makes it a compile error (C++11 `final`) instead
    of a silent difference; deriving without touching the NDF, and deriving from djb::radial with an NDF of one's own, stay legal."""
    bad = '#include "dj_brdf.h"\nstruct my : djb::ggx { float p22_radial(float r) const { return r; } };\nint main() { my m; return 0; }\n'
    r, _ = build(tmp_path, bad)
    assert r.returncode != 0 and ("final" in r.stderr), r.stderr[-1500:]
    ok = ('#include "dj_brdf.h"\nstruct my : djb::ggx { float twice(const djb::vec3 &i, const djb::vec3 &o) const { return 2 * pdf(i, o); } };\n'
          'int main() { my m; return m.twice(djb::vec3(0, 0, 1), djb::vec3(0, 0, 1)) > 0 ? 0 : 1; }\n')
    r, exe = build(tmp_path, ok)
    assert r.returncode == 0, r.stderr
    out = subprocess.run([str(exe)], env=dict(os.environ, DJB_DEVICE="cpu", DJB_QUIET="1"), capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr


def test_member_signatures_are_the_reference_ones():
    """tests/api/api_signature_probe.cpp asserts the exact type of every public member (return type, parameters, const, static), the
    abstract / noncopyable / convertible properties of the classes, and calls everything once with its default arguments omitted.  It
    compiles against the reference (which proves the assertions, where the reference is mounted) and must compile against the facade."""
    src = os.path.join(ROOT, "tests", "api", "api_signature_probe.cpp")
    r = subprocess.run(["g++", "-std=c++11", "-fsyntax-only", "-DNVERBOSE", "-Wall", "-I", os.path.join(ROOT, "include"), src], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]
    if os.path.exists("/root/reference/dj_brdf.h"):
        r = subprocess.run(["g++", "-std=c++11", "-fsyntax-only", "-DNVERBOSE", "-w", "-I", "/root/reference", src], capture_output=True, text=True)
        assert r.returncode == 0, "the probe does not compile against the reference itself:\n" + r.stderr[-4000:]


THROWING_BATCH = r"""
#include <stdexcept>
#include <cstdio>
#include <cmath>
#include <vector>
#include "dj_brdf.h"
// a radial NDF (the reference's extension point, dj_brdf.h:301-324) whose p22_radial throws beyond a slope limit
class throwing_ndf : public djb::radial {
public:
	throwing_ndf(float limit) : m_limit(limit) {}
	bool supports_smith_vndf_sampling() const { return false; }
	float p22_radial(float r_sqr) const { if (r_sqr > m_limit) throw std::out_of_range("slope beyond the table"); return (float)(1.0 / (M_PI * (1.0 + (double)r_sqr) * (1.0 + (double)r_sqr))); }
	float sigma_std_radial(float c) const { return 0.5f * (1.0f + c); }
	float cdf_radial(float r) const { return r * r / (1.0f + r * r); }
	float qf_radial(float u) const { return (float)std::sqrt((double)u / (1.0 - (double)u)); }
private:
	float m_limit;
};
int main()
{
	throwing_ndf t(0.5f);
	const int n = 400;                       // above DJB_SCALAR_HOST_MAX: the batch entry point, not the one-pair path
	std::vector<djb::vec3> i(n), o(n), out(n);
	for (int k = 0; k < n; ++k) {
		const float a = 0.02f + 1.5f * k / n;            // half vectors from the normal to grazing: some slopes exceed the limit
		i[k] = djb::vec3(std::sin(a), 0, std::cos(a)); o[k] = djb::vec3(std::sin(a) * 0.5f, 0.1f, std::sqrt(1 - 0.25f * std::sin(a) * std::sin(a) - 0.01f));
	}
	int caught = 0;
	try { t.eval((size_t)n, &i[0], &o[0], &out[0]); } catch (const std::out_of_range &e) { ++caught; printf("eval batch: out_of_range: %s\n", e.what()); }
	try { t.evalp((size_t)n, &i[0], &o[0], &out[0]); } catch (const std::out_of_range &e) { ++caught; printf("evalp batch: out_of_range: %s\n", e.what()); }
	// the object stays usable, and nothing is left behind to surface on an unrelated call
	const djb::vec3 v = t.eval(djb::vec3(0, 0, 1), djb::vec3(0, 0, 1));
	printf("caught=%d after=%d\n", caught, (int)(v.x == v.x));
	return 0;
}
"""


def test_user_ndf_exception_in_a_batch_reaches_the_caller(tmp_path):
    """microfacet::host_eval_batch (user NDF + library Fresnel) must re-throw what the user's callback threw during the
    batch, at that call -- not return NaNs and surface the exception on a later, unrelated call."""
    r, exe = build(tmp_path, THROWING_BATCH, ["-DNVERBOSE"])
    assert r.returncode == 0, r.stderr
    out = subprocess.run([str(exe)], env=dict(os.environ, DJB_DEVICE="cpu", DJB_QUIET="1"), capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert "eval batch: out_of_range: slope beyond the table" in out.stdout and "evalp batch: out_of_range" in out.stdout
    assert "caught=2 after=1" in out.stdout
