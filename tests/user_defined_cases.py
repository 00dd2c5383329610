"""Checks of the reference's EXTENSION POINTS on the product path, shared by the CPU suite (tests/test_cpu_path.py, host
execution path) and the GPU suite (tests/test_gpu_custom.py): a BRDF the caller defines (dj_brdf.h:74-109) is fitted by
sampling its eval() on the host at the fit's query directions (djb_fit_query_dirs) and running the fit kernels on the
samples (djb_brdf_create_tabular_from_samples / ..._anisotropic_from_samples).

The user's lobes are the fixtures of oracle/ref_shim.cpp (user_phong, user_ward); here the caller-side eval() is
answered by the oracle's restatement of them, and everything the PRODUCT computes from those samples -- the power
iteration, the quadratures, the Fresnel ratios, cdf / qf, both moment fits -- is compared bit for bit with
tests/golden/custom.npz (= the REAL reference running the same classes) and with the oracle's own fit."""
import os

import numpy as np

from dj_brdf_amd import djb
from golden_cases import CUSTOM_ANISO, CUSTOM_FITS, CUSTOM_LOBES

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def same(a, b):
    a, b = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
    return a.shape == b.shape and bool(((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))).all())


def make_user_brdf(oracle, lobe, ctx, calls=None):
    handle = oracle.custom(*lobe)

    class lobe_brdf(djb.user_brdf):                       # what a user writes: derive, override eval
        def eval(self, i, o, user_param=None):
            if calls is not None:
                calls.append((np.array(i), np.array(o)))
            return oracle.eval(handle, i, o)
    return lobe_brdf(ctx=ctx), handle


def check_user_defined_fits(ctx, oracle, name):
    g = np.load(os.path.join(G, "custom.npz"))
    lobe = CUSTOM_LOBES[name]
    calls = []
    b, ob = make_user_brdf(oracle, lobe, ctx, calls)
    # the base-class operators of a user-derived object (dj_brdf.h:795-845)
    i, o, u1, u2, h, d = (g[k] for k in ("i", "o", "u1", "u2", "h", "d"))
    assert same(b.eval(i, o), g[f"{name}_eval"]) and same(b.evalp(i, o), g[f"{name}_evalp"]) and same(b.pdf(i, o), g[f"{name}_pdf"])
    assert same(b.eval_hd(h, d), g[f"{name}_eval_hd"]) and same(b.evalp_hd(h, d), g[f"{name}_evalp_hd"])
    assert same(b.sample(u1, u2, o), g[f"{name}_sample"])
    w, si, pdf = b.evalp_is(u1, u2, o)
    assert same(w, g[f"{name}_is_w"]) and same(si, g[f"{name}_is_i"]) and same(pdf, g[f"{name}_is_pdf"])
    # isotropic fits
    for res, shadow in CUSTOM_FITS:
        del calls[:]
        t = djb.tabular(b, res, shadow, ctx=ctx)
        # eval was called where the reference calls it, in its order: res-1 back-scatter pairs, then i = (0, 0, 1)
        (qi, qo), = calls
        cnt = res - 1
        assert qi.shape[0] <= cnt * (cnt + 2) and same(qi[:cnt], qo[:cnt]) and (qi[cnt:] == np.float32([0, 0, 1])).all()
        want = oracle.tabular_tables(oracle.tabular(ob, res, shadow))
        for k, v in (("p22", t.get_p22v()), ("sigma", t.get_sigmav()), ("cdf", t.get_cdfv()), ("qf", t.get_qfv()),
                     ("fresnel", t.get_fresnel().get_points())):
            assert same(v, g[f"{name}_fit{res}_{k}"]), (name, res, k)
            assert same(v, want[k]), (name, res, k)
        ab = np.float32(djb.tabular.fit_beckmann_parameters(t).get_ellipse()[0])
        ag = np.float32(djb.tabular.fit_ggx_parameters(t).get_ellipse()[0])
        assert (ab, ag) == (g[f"{name}_fit{res}_alpha_beckmann"][0], g[f"{name}_fit{res}_alpha_ggx"][0])
        assert "%.3f %.3f" % (ab, ag) == "%.3f %.3f" % (want["alpha_beckmann"], want["alpha_ggx"])
        # the fitted object is an ordinary resident tabular BRDF
        assert same(t.eval(i, o), oracle.eval(oracle.tabular(ob, res, shadow), i, o))
    # the same fit from samples handed over directly (the C ABI's form)
    res, shadow = CUSTOM_FITS[0]
    qi, qo = djb.fit_query_dirs(res)
    ok = ~np.isnan(qo[:, 0])
    rgb = np.full((qi.shape[0], 3), np.nan, np.float32)          # skipped slots: any value
    rgb[ok] = oracle.eval(ob, qi[ok], qo[ok])
    t2 = djb.tabular.from_samples(res, rgb, shadow, ctx=ctx)
    assert same(t2.get_p22v(), g[f"{name}_fit{res}_p22"]) and same(t2.get_fresnel().get_points(), g[f"{name}_fit{res}_fresnel"])
    # anisotropic fit
    elev, azim = CUSTOM_ANISO
    ta = djb.tabular_anisotropic(b, elev, azim, True, ctx=ctx)
    assert same(ta.get_p22v()[0], g[f"{name}_aniso_p22"]) and same(ta.get_sigmav()[0], g[f"{name}_aniso_sigma"])
    assert same(ta.get_fresnel().get_points(), g[f"{name}_aniso_fresnel"])
    fb = np.array(djb.tabular_anisotropic.fit_beckmann_parameters(ta).get_pdfparams(), np.float32)
    fg = np.array(djb.tabular_anisotropic.fit_ggx_parameters(ta).get_pdfparams(), np.float32)
    assert same(fb, g[f"{name}_aniso_fit_beckmann"]) and same(fg, g[f"{name}_aniso_fit_ggx"])


def check_lambert_source(ctx, oracle):
    """tabular / tabular_anisotropic of the library's own Lambertian (eval with user_param == NULL: reflectance 1) against the oracle --
    and the same fit from samples of its eval(): three ways to the same tables.  (Round 5: the host path read the microfacet
    standard params' normal (0, 0, 1) as the reflectance and fitted a blue-only source; no test had a Lambertian source.)"""
    lam, olam = djb.lambert(ctx=ctx), oracle.lambert()
    for res in (32, 90):
        t = djb.tabular(lam, res, True, ctx=ctx)
        want = oracle.tabular_tables(oracle.tabular(olam, res, True))
        qi, qo = djb.fit_query_dirs(res)
        ok = ~np.isnan(qo[:, 0])
        rgb = np.zeros((qi.shape[0], 3), np.float32)
        rgb[ok] = lam.eval(qi[ok], qo[ok])
        t2 = djb.tabular.from_samples(res, rgb, True, ctx=ctx)
        for k, v, v2 in (("p22", t.get_p22v(), t2.get_p22v()), ("sigma", t.get_sigmav(), t2.get_sigmav()), ("cdf", t.get_cdfv(), t2.get_cdfv()),
                         ("qf", t.get_qfv(), t2.get_qfv()), ("fresnel", t.get_fresnel().get_points(), t2.get_fresnel().get_points())):
            assert same(v, want[k]) and same(v2, want[k]), (res, k)
    ta = djb.tabular_anisotropic(lam, 9, 16, True, ctx=ctx)
    wa = oracle.aniso_tables(oracle.tabular_anisotropic(olam, 9, 16, True))
    assert same(ta.get_p22v()[0], wa["p22"]) and same(ta.get_sigmav()[0], wa["sigma"]) and same(ta.get_fresnel().get_points(), wa["fresnel"])


def check_sample_count_errors(ctx):
    import pytest
    qi, qo = djb.fit_query_dirs(20)
    assert qi.shape == (19 * 21, 3) and np.isnan(qo[:, 0]).any() and not np.isnan(qo[:19]).any()
    ai, ao = djb.fit_aniso_query_dirs(9, 16)
    assert ai.shape == (8 * 16 + 8 * 9, 3)
    with pytest.raises(djb.exc) as e:
        djb.tabular.from_samples(20, np.zeros((7, 3), np.float32), ctx=ctx)
    assert e.value.status_name == "DJB_ERR_INVALID_ARGUMENT" and "samples" in str(e.value)
    with pytest.raises(djb.exc):
        djb.tabular_anisotropic.from_samples(9, 16, np.zeros((7, 3), np.float32), ctx=ctx)
    with pytest.raises(djb.exc) as e:
        djb.fit_query_dirs(2)
    assert "Invalid Resolution" in str(e.value)
