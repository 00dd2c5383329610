"""Conformance of the C++ facade (include/djb_hip.hpp via include/dj_brdf.h), class by class.

oracle/ref_shim.cpp -- the extern "C" wrapper that exposes every public class of the REAL reference to the
test-suite -- is compiled a second time, unchanged apart from -DDJB_FACADE_SHIM, against this
repository's dj_brdf.h (oracle/Makefile `facade`).  The resulting library has the same ref_* entry
points, but every call constructs the facade's djb:: objects on the GPU context; constructors and fits
run on the GPU, and each element is one scalar facade call.  Scalar calls have two execution paths and the
whole module runs once on each:
  * "twin"   (default): answered on the calling thread from the host twin of the GPU object (csrc/djb_host.hip);
  * "device" (DJB_OPT_SCALAR_ON_DEVICE): a launch + a PCIe round trip per call -- the kernels' operator path.
Here the library is driven exactly like the real reference in tests/test_oracle_vs_ref.py and compared
with the CPU oracle (which is itself pinned bit-exact to the reference)."""
import os

import numpy as np
import pytest

import oraclelib
from dj_brdf_amd import synth
from golden_cases import PARAM_CASES, _FRESNELS

FRESNEL_CASES = [("ideal",)] + list(_FRESNELS)

pytestmark = pytest.mark.gpu
SHIM = os.path.join(oraclelib.ORACLE_DIR, "_facade", "libdjb_facade_shim.so")
N = 192   # every element is one scalar facade call


@pytest.fixture(scope="module", params=["twin", "device"])
def facade(gpu_ctx, request):
    if not os.path.exists(SHIM):
        pytest.skip("oracle/_facade/libdjb_facade_shim.so not built (make -C oracle facade)")
    lib = oraclelib.CheckerLib(SHIM, "ref_")
    assert lib._fn("facade_scalar_on_device")(1 if request.param == "device" else 0) == 0
    yield lib
    lib._fn("facade_scalar_on_device")(0)


@pytest.fixture(scope="module")
def inputs():
    return (synth.directions_aos(N, synth.SEED_I, start=777), synth.directions_aos(N, synth.SEED_O, start=777),
            synth.uniforms(N, synth.SEED_U1, start=777), synth.uniforms(N, synth.SEED_U2, start=777))


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def close(tag, got, want, rtol=0.0):
    """rtol: the contract at that call site (north_star 1e-5 and up); asserted: identical bits, which is
    what the GPU path delivers everywhere -- the figure only classifies a failure."""
    got, want = np.atleast_1d(np.asarray(got, np.float32)), np.atleast_1d(np.asarray(want, np.float32))
    assert got.shape == want.shape, tag
    same = (np.isnan(got) & np.isnan(want)) | (bits(got) == bits(want))
    if not same.all():
        rel = np.nanmax(np.abs(got.astype(np.float64) - want) / np.maximum(np.abs(want), 1e-30))
        raise AssertionError(f"{tag}: {np.mean(~same):.2e} of values not bit-identical; max rel {rel:.2e} (contract {rtol or 1e-5})")


@pytest.mark.parametrize("ndf", ["ggx", "beckmann"])
def test_microfacet_classes(facade, oracle, inputs, ndf):
    i, o, u1, u2 = inputs
    for fres in FRESNEL_CASES:
        for shadow in (True, False):
            f, b = facade.microfacet(ndf, fres, shadow), oracle.microfacet(ndf, fres, shadow)
            for p in PARAM_CASES[:4]:
                for op in ("eval", "evalp", "pdf"):
                    close(f"{ndf}/{fres[0]}/{p}/{op}", facade.eval(f, i, o, p, op), oracle.eval(b, i, o, p, op))
            p = PARAM_CASES[2]
            close("sample", facade.sample(f, u1, u2, o, p), oracle.sample(b, u1, u2, o, p))
            fw, fi, fp = facade.evalp_is(f, u1, u2, o, p)
            ow, oi, op_ = oracle.evalp_is(b, u1, u2, o, p)
            close("evalp_is i", fi, oi); close("evalp_is pdf", fp, op_); close("evalp_is weight", fw, ow)
            c = np.clip(o[:, 2], 0, 1)
            close("fresnel()", facade.fresnel_eval(f, c), oracle.fresnel_eval(b, c))
            facade.destroy(f)


def test_user_defined_classes(facade, oracle, inputs):
    """Classes DERIVED BY THE USER from djb::brdf and djb::fresnel::impl (ref_shim.cpp: user_phong, user_ward, user_lazanyi),
    compiled against include/dj_brdf.h: base-class operators (dj_brdf.h:795-845), eval_hd / evalp_hd, fits of the user's lobes
    (their eval() sampled on the host, the fit on the GPU), and the library's microfacet BRDFs holding the user's Fresnel term
    (D G on the GPU / host twin, F on the host)."""
    from golden_cases import CUSTOM_ANISO, CUSTOM_FITS, CUSTOM_FRESNEL, CUSTOM_FRESNEL_FIT, CUSTOM_LOBES, CUSTOM_PARAMS
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "custom.npz"))
    i, o, u1, u2 = inputs
    h, d = oracle.io_to_hd(i, o)
    for name, lobe in CUSTOM_LOBES.items():
        f, b = facade.custom(*lobe), oracle.custom(*lobe)
        for op in ("eval", "evalp", "pdf"):
            close(f"{name}/{op}", facade.eval(f, i, o, None, op), oracle.eval(b, i, o, None, op))
        for op in ("eval_hd", "evalp_hd"):
            close(f"{name}/{op}", facade.eval(f, h, d, None, op), oracle.eval(b, h, d, None, op))
        close(f"{name}/sample", facade.sample(f, u1, u2, o), oracle.sample(b, u1, u2, o))
        for tag, x, y in zip(("w", "i", "pdf"), facade.evalp_is(f, u1, u2, o), oracle.evalp_is(b, u1, u2, o)):
            close(f"{name}/evalp_is {tag}", x, y)
        for res, shadow in CUSTOM_FITS:
            got = facade.tabular_tables(facade.tabular(f, res, shadow))
            for k, v in got.items():
                close(f"{name}/fit{res}/{k}", np.atleast_1d(v), g[f"{name}_fit{res}_{k}"])
        got = facade.aniso_tables(facade.tabular_anisotropic(f, *CUSTOM_ANISO))
        for k, v in got.items():
            close(f"{name}/aniso/{k}", v, g[f"{name}_aniso_{k}"])
        facade.destroy(f)
    for ndf in ("ggx", "beckmann"):
        for shadow in (True, False):
            f, b = facade.microfacet(ndf, CUSTOM_FRESNEL, shadow), oracle.microfacet(ndf, CUSTOM_FRESNEL, shadow)
            for p in (CUSTOM_PARAMS, None):
                for op in ("eval", "evalp", "pdf"):
                    close(f"{ndf}/user fresnel/{op}", facade.eval(f, i, o, p, op), oracle.eval(b, i, o, p, op))
                for op in ("eval_hd", "evalp_hd"):
                    close(f"{ndf}/user fresnel/{op}", facade.eval(f, h, d, p, op), oracle.eval(b, h, d, p, op))
                for tag, x, y in zip(("w", "i", "pdf"), facade.evalp_is(f, u1, u2, o, p), oracle.evalp_is(b, u1, u2, o, p)):
                    close(f"{ndf}/user fresnel/evalp_is {tag}", x, y)
            c = np.clip(o[:, 2], 0, 1)
            close("user fresnel()", facade.fresnel_eval(f, c), oracle.fresnel_eval(b, c))
            facade.destroy(f)
        # eval_hd / evalp_hd of the library's own classes: evalp_hd is eval * cos (dj_brdf.h:808-814), not evalp
        f, b = facade.microfacet(ndf, ("schlick", 0.9, 0.5, 0.1), True), oracle.microfacet(ndf, ("schlick", 0.9, 0.5, 0.1), True)
        for op in ("eval_hd", "evalp_hd"):
            close(f"{ndf}/schlick/{op}", facade.eval(f, h, d, CUSTOM_PARAMS, op), oracle.eval(b, h, d, CUSTOM_PARAMS, op))
        facade.destroy(f)
        f = facade.microfacet(ndf, CUSTOM_FRESNEL, True)
        got = facade.tabular_tables(facade.tabular(f, CUSTOM_FRESNEL_FIT, True))
        for k, v in got.items():
            close(f"{ndf}/user fresnel/fit/{k}", np.atleast_1d(v), g[f"{ndf}_fit_{k}"])
        facade.destroy(f)


def test_user_defined_ndf_classes(facade):
    """NDF classes DERIVED BY THE USER from djb::radial (public virtuals p22_radial / sigma_std_radial / cdf_radial / qf_radial,
    dj_brdf.h:301-324) and from djb::microfacet (protected p22_std / sigma_std / sample_vp22_std_nmap, dj_brdf.h:283-295), compiled
    against include/dj_brdf.h: the objects live on the library's host path with the user's functions as callbacks, everything around
    the NDF is the library's per-unit code.  Against tests/golden/custom.npz = the REAL reference running the same classes
    (ref_shim.cpp: user_student, user_separable), with a library Fresnel term and with the user's; fits of them run on the GPU."""
    from golden_cases import CUSTOM_ANISO, CUSTOM_FRESNEL, CUSTOM_NDFS, CUSTOM_NDF_PARAMS, CUSTOM_NDF_QUERIES
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "custom.npz"))
    M = 128
    i, o, u1, u2, h, d = (g[k][:M] for k in ("i", "o", "u1", "u2", "h", "d"))
    for ndf in CUSTOM_NDFS:
        for fk, fres in (("ideal", ("ideal",)), ("schlick", ("schlick", 0.9, 0.5, 0.1)), ("user", CUSTOM_FRESNEL)):
            for shadow in (True, False):
                f = facade.microfacet(ndf, fres, shadow)
                for pk, par in enumerate(CUSTOM_NDF_PARAMS):
                    tag = f"{ndf}_{fk}{int(shadow)}_p{pk}"
                    for op in ("eval", "evalp", "pdf"):
                        close(f"{tag}/{op}", facade.eval(f, i, o, par, op), g[f"{tag}_{op}"])
                    close(f"{tag}/evalp_hd", facade.eval(f, h, d, par, "evalp_hd"), g[f"{tag}_evalp_hd"])
                    close(f"{tag}/sample", facade.sample(f, u1, u2, o, par), g[f"{tag}_sample"])
                    for t, x in zip(("is_w", "is_i", "is_pdf"), facade.evalp_is(f, u1, u2, o, par)):
                        close(f"{tag}/{t}", x, g[f"{tag}_{t}"])
                    if fk == "ideal" and shadow:
                        args = {"h": i, "i": i, "o": o}
                        for q, sig in CUSTOM_NDF_QUERIES:
                            close(f"{tag}/{q}", facade.microfacet_query(f, q, *[args[c] for c in sig], params=par), g[f"{tag}_{q}"])
                facade.destroy(f)
        f = facade.microfacet(ndf, ("ideal",), True)
        for k, v in facade.tabular_tables(facade.tabular(f, 40, True)).items():
            close(f"{ndf}/fit/{k}", np.atleast_1d(v), g[f"{ndf}_fit_{k}"])
        for k, v in facade.aniso_tables(facade.tabular_anisotropic(f, *CUSTOM_ANISO)).items():
            close(f"{ndf}/aniso/{k}", v, g[f"{ndf}_aniso_{k}"])
        facade.destroy(f)


def test_params_vec3_and_helpers(facade, oracle, inputs):
    i, o, _, _ = inputs
    for p in PARAM_CASES:
        want = oracle.params_get(p)
        got = facade.params_get(p)
        ok = np.isnan(want) | (bits(got) == bits(want))
        assert ok.all(), (p, got, want)
        close(f"lrep roundtrip {p}", facade.params_lrep_roundtrip(p), oracle.params_lrep_roundtrip(p))
    h, d = facade.io_to_hd(i, o); wh, wd = oracle.io_to_hd(i, o)
    close("io_to_hd h", h, wh); close("io_to_hd d", d, wd)
    bi, bo = facade.hd_to_io(wh, wd); wi, wo = oracle.hd_to_io(wh, wd)
    close("hd_to_io i", bi, wi, 1e-5); close("hd_to_io o", bo, wo, 2e-5)
    th = np.linspace(0, np.pi, N).astype(np.float32); ph = np.linspace(-7, 7, N).astype(np.float32)
    close("vec3(theta, phi)", facade.vec3_angles(th, ph), oracle.vec3_angles(th, ph))
    x = np.linspace(0.3, 6, N).astype(np.float32); f0 = np.linspace(0, 0.999, N).astype(np.float32)
    close("ior_to_f0", facade.ior_f0(0, x), oracle.ior_f0(0, x)); close("f0_to_ior", facade.ior_f0(1, f0), oracle.ior_f0(1, f0))


def test_queries(facade, oracle, inputs):
    i, o, u1, u2 = inputs
    h = synth.directions_aos(N, 99)
    for ndf in ("ggx", "beckmann"):
        f, b = facade.microfacet(ndf), oracle.microfacet(ndf)
        p = PARAM_CASES[3]
        for which, args in (("ndf", (h,)), ("gaf", (h, i, o)), ("g1", (h, o)), ("sigma", (o,)), ("vndf", (h, o))):
            close(f"{ndf} {which}", facade.microfacet_query(f, which, *args, params=p), oracle.microfacet_query(b, which, *args, params=p), 1e-5)
        c = np.clip(o[:, 2], 1e-3, 1).astype(np.float32); s = np.sqrt(1 - c.astype(np.float64) ** 2).astype(np.float32)
        u = np.clip(u1, 1e-3, 1 - 1e-3)
        for which, args in (("p22_radial", (u * 9,)), ("sigma_std_radial", (c,)), ("cdf_radial", (u * 5,)), ("qf_radial", (u,)),
                            ("qf3_radial", (u, u * 3 - 1))):
            close(f"{ndf} {which}", facade.radial_query(f, which, *args), oracle.radial_query(b, which, *args), 2e-5)
        facade.destroy(f)


def test_lambert_merl_utia_models(facade, oracle, inputs, tmp_path):
    i, o, u1, u2 = inputs
    fl, ol = facade.lambert(), oracle.lambert()
    for p in (None, ("lambert", 0.5, 0.25, 0.9)):
        for op in ("eval", "evalp", "pdf"):
            close(f"lambert {p} {op}", facade.eval(fl, i, o, p, op), oracle.eval(ol, i, o, p, op))
    close("lambert sample", facade.sample(fl, u1, u2, o), oracle.sample(ol, u1, u2, o), 2e-5)
    tab = synth.merl_table_hashed()
    path = str(tmp_path / "m.binary"); synth.write_merl_binary(path, tab)
    fm, om = facade.merl(path), oracle.merl(path)
    for op in ("eval", "evalp", "pdf"):
        close(f"merl {op}", facade.eval(fm, i, o, None, op), oracle.eval(om, i, o, None, op))
    with pytest.raises(RuntimeError) as e:
        facade.merl(str(tmp_path / "missing.binary"))
    assert "Failed to open" in str(e.value)
    ut = np.random.default_rng(5).uniform(-5, 120, 3 * 288 * 288); up = str(tmp_path / "u.bin"); ut.tofile(up)
    close("utia eval", facade.eval(facade.utia(up), i, o), oracle.eval(oracle.utia(up), i, o), 1e-5)
    for kind in ("sgd", "abc"):
        f, b = getattr(facade, kind)("gold-metallic-paint"), getattr(oracle, kind)("gold-metallic-paint")
        close(f"{kind} eval", facade.eval(f, i, o), oracle.eval(b, i, o), 1e-5)
        close(f"{kind} ndf", facade.model_query(f, "ndf", i), oracle.model_query(b, "ndf", i), 1e-5)
        close(f"{kind} gaf", facade.model_query(f, "gaf", i, o, i), oracle.model_query(b, "gaf", i, o, i), 1e-5)
        cc = np.zeros_like(i); cc[:, 0] = np.clip(i[:, 2], 0, 1)
        close(f"{kind} get_fresnel().eval", facade.model_query(f, "get_fresnel", cc), oracle.model_query(b, "fresnel", cc))
        if kind == "sgd":
            close("sgd g1", facade.model_query(f, "g1", o), oracle.model_query(b, "g1", o), 1e-5)
        with pytest.raises(RuntimeError):
            getattr(facade, kind)("no-such-material")


def test_tabular_and_lrep(facade, oracle, inputs):
    i, o, u1, u2 = inputs
    ft, ot = facade.tabular(facade.microfacet("ggx"), 64, True), oracle.tabular(oracle.microfacet("ggx"), 64, True)
    for k, v in oracle.tabular_tables(ot).items():
        got = facade.tabular_tables(ft)[k]
        close(f"tabular table {k}", got, v)
    close("tabular eval", facade.eval(ft, i, o), oracle.eval(ot, i, o), 1e-4)
    from golden_cases import lrep_cases
    for op, a, b, x, y in lrep_cases():
        close(f"lrep {op}", facade.lrep_op(op, a, b, x, y), oracle.lrep_op(op, a, b, x, y))


def test_anisotropic_and_lean(facade, oracle, inputs):
    i, o, u1, u2 = inputs
    from golden_cases import LEAN_BASE, LEAN_SCALE, lean_moments, lean_texels
    # dj_beckmannconductor's per-hit path through beckmann::lrep and per-call microfacet::params
    lean = lean_moments(N)
    for ndf in ("beckmann", "ggx"):
        f, b = facade.microfacet(ndf, ("schlick", 1.0, 0.71, 0.29), True), oracle.microfacet(ndf, ("schlick", 1.0, 0.71, 0.29), True)
        for scale, filtering, biased in ((LEAN_SCALE, True, False), (2.0, False, False), (0.5, True, True)):
            tex = lean_texels(lean, biased)
            for op in ("evalp", "pdf"):
                got, gpp = facade.eval_lean(f, i, o, LEAN_BASE, scale, tex, op, filtering=filtering, biased=biased)
                want, wpp = oracle.eval_lean(b, i, o, LEAN_BASE, scale, tex, op, filtering=filtering, biased=biased)
                close(f"{ndf} lean params", gpp, wpp); close(f"{ndf} lean {op}", got, want)
    # tabular_anisotropic on a small grid: tables, fits, two-level sampling queries, eval
    ft = facade.tabular_anisotropic(facade.microfacet("ggx"), 12, 16, True)
    ot = oracle.tabular_anisotropic(oracle.microfacet("ggx"), 12, 16, True)
    want = oracle.aniso_tables(ot)
    for k, v in facade.aniso_tables(ft).items():
        close(f"aniso table {k}", v, want[k])
    phi = (u1 * 2 * np.pi).astype(np.float32); th = (u2 * 1.5).astype(np.float32)
    for which, args in (("pdf1", (phi,)), ("cdf1", (phi,)), ("qf1", (u1,)), ("pdf2", (th, phi)), ("cdf2", (th, phi)), ("qf2", (u2, phi))):
        close(f"aniso {which}", facade.aniso_query(ft, which, *args), oracle.aniso_query(ot, which, *args), 2e-4)
    close("aniso eval", facade.eval(ft, i, o), oracle.eval(ot, i, o), 2e-4)
