"""sgd / abc analytic models on the HIP path: eval/evalp vs the golden vectors
of the real reference and vs the oracle, default cosine sampling, and the tabular(model, 90) fit
that the dj_sgd / dj_abc plugins run at load time."""
import os

import numpy as np
import pytest

from dj_brdf_amd import djb, param_tables, synth
from golden_cases import MODEL_MATERIALS
from test_gpu_parity import assert_close

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("kind", ["sgd", "abc"])
def test_models_golden(gpu_ctx, kind):
    g = np.load(os.path.join(G, "models.npz"))
    for name in MODEL_MATERIALS:
        b = getattr(djb, kind)(name, ctx=gpu_ctx)
        ex = assert_close(f"{kind}/{name}/eval", b.eval(g["i"], g["o"]), g[f"{kind}_{name}_eval"])
        assert ex > 0.999, f"{kind}/{name}: only {ex:.5f} bit-identical"
        assert_close(f"{kind}/{name}/evalp", b.evalp(g["i"], g["o"]), g[f"{kind}_{name}_evalp"])


@pytest.mark.parametrize("kind", ["sgd", "abc"])
def test_models_all_materials_vs_oracle(gpu_ctx, oracle, kind):
    n = 1 << 14
    i = synth.directions_aos(n, synth.SEED_I, 31); o = synth.directions_aos(n, synth.SEED_O, 31)
    u1 = synth.uniforms(n, synth.SEED_U1); u2 = synth.uniforms(n, synth.SEED_U2)
    below = i.copy(); below[: n // 8, 2] *= -1                      # i below the horizon -> 0
    for name in synth.MERL_NAMES:
        b, ob = getattr(djb, kind)(name, ctx=gpu_ctx), getattr(oracle, kind)(name)
        assert_close(f"{kind}/{name}", b.eval(below, o), oracle.eval(ob, below, o))
    b, ob = getattr(djb, kind)("pearl-paint", ctx=gpu_ctx), getattr(oracle, kind)("pearl-paint")
    assert_close("pdf", b.pdf(i, o), oracle.eval(ob, i, o, None, "pdf"))
    assert_close("sample", b.sample(u1, u2, o), oracle.sample(ob, u1, u2, o), 2e-5)
    w, si, pdf = b.evalp_is(u1, u2, o)
    ww, wi, wpdf = oracle.evalp_is(ob, u1, u2, o)
    assert_close("is pdf", pdf, wpdf, 2e-5); assert_close("is w", w, ww, 5e-5)
    # the same object from an explicit parameter row
    row = param_tables.sgd_params("pearl-paint") if kind == "sgd" else param_tables.abc_params("pearl-paint")
    b2 = getattr(djb, kind).from_params(row, ctx=gpu_ctx)
    assert np.array_equal(b2.eval(i, o).view(np.uint32), b.eval(i, o).view(np.uint32))


@pytest.mark.parametrize("kind", ["sgd", "abc"])
def test_model_member_queries(gpu_ctx, oracle, kind):
    """sgd::{ndf, gaf, g1, fresnel} / abc::{ndf, gaf, fresnel} (dj_brdf.h:505-509, 530-533)."""
    n = 1 << 14
    h = synth.directions_aos(n, 5); i = synth.directions_aos(n, 6); o = synth.directions_aos(n, 7)
    c = np.clip(h[:, 2], 0, 1)
    cc = np.zeros((n, 3), np.float32); cc[:, 0] = c
    for name in ("gold-metallic-paint", "alum-bronze", "beige-fabric", "pearl-paint"):
        b, ob = getattr(djb, kind)(name, ctx=gpu_ctx), getattr(oracle, kind)(name)
        ex = assert_close(f"{kind}/{name} ndf", b.ndf(h), oracle.model_query(ob, "ndf", h), 1e-5)
        assert ex > 0.99
        assert_close(f"{kind}/{name} fresnel", b.fresnel(c), oracle.model_query(ob, "fresnel", cc), 1e-5)
        want = oracle.model_query(ob, "gaf", h, i, o)
        if kind == "sgd":
            assert_close(f"{name} gaf", b.gaf(h, i, o), want, 1e-5)
            assert_close(f"{name} g1", b.g1(i), oracle.model_query(ob, "g1", i), 1e-5)
        else:
            assert_close(f"{name} gaf", b.gaf(h, i, o), want[:, 0], 1e-5)
            with pytest.raises(djb.exc):
                djb.microfacet._query(b, 50, i)          # abc has no g1
    with pytest.raises(djb.exc):
        djb.microfacet._query(djb.ggx(ctx=gpu_ctx), 48, h)   # model queries need sgd / abc


def test_unknown_material_raises(gpu_ctx):
    for kind, msg in (("sgd", "No SGD parameters for nope"), ("abc", "No ABC parameters for nope")):
        with pytest.raises(djb.exc) as e:
            getattr(djb, kind)("nope", ctx=gpu_ctx)
        assert e.value.status_name == "DJB_ERR_UNKNOWN_MATERIAL" and msg in str(e.value)
    djb.sgd("fabric-beige", ctx=gpu_ctx)            # SGD alias (otherName), dj_brdf.h:3440-3441


@pytest.mark.parametrize("kind", ["sgd", "abc"])
def test_plugin_load_time_fit(gpu_ctx, oracle, kind):
    """djb::tabular(model, 90): nmap-sampled tabulated lobe used by dj_abc / dj_sgd for sample()/pdf()."""
    g = np.load(os.path.join(G, "models.npz"))
    t = djb.tabular(getattr(djb, kind)(MODEL_MATERIALS[0], ctx=gpu_ctx), 90, True, ctx=gpu_ctx)
    got = {"p22": t.get_p22v(), "sigma": t.get_sigmav(), "cdf": t.get_cdfv(), "qf": t.get_qfv(),
           "fresnel": t.get_fresnel().get_points()}
    for k, v in got.items():
        assert_close(f"{kind}/fit/{k}", v, g[f"{kind}_fit_{k}"], rtol=2e-5)
    ab = djb.tabular.fit_beckmann_parameters(t).get_ellipse()[0]
    ag = djb.tabular.fit_ggx_parameters(t).get_ellipse()[0]
    assert (np.float32(ab), np.float32(ag)) == (g[f"{kind}_fit_alpha_beckmann"][0], g[f"{kind}_fit_alpha_ggx"][0])
    n = 4096
    o = synth.directions_aos(n, synth.SEED_O); u1 = synth.uniforms(n, 1); u2 = synth.uniforms(n, 2)
    ot = oracle.tabular(getattr(oracle, kind)(MODEL_MATERIALS[0]), 90, True)
    s = t.sample(u1, u2, o)
    assert_close("sample", s, oracle.sample(ot, u1, u2, o))
    assert_close("pdf", t.pdf(s, o), oracle.eval(ot, s, o, None, "pdf"), 1e-4)


@pytest.mark.parametrize("kind", ["sgd", "abc"])
def test_model_fast_tier_selftest(gpu_ctx, kind):
    """The decided fast tier of the models' fp64 chains (csrc/djb_fast_models.inc) against the exact chains on the device:
    every published row, 2^22 generated polar cosines each (uniform, wall-hugging, grazing, near the normal) -- no value may differ."""
    worst = 0.0
    for name in synth.MERL_NAMES:
        b = getattr(djb, kind)(name, ctx=gpu_ctx)
        r = djb.selftest_model_fast(b, 1 << 22, seed=7, ctx=gpu_ctx)
        assert r["g1_mismatch"] == 0 and r["ndf_mismatch"] == 0, (name, r)
        assert r["ndf"] == 3 << 22 and (kind == "abc" or r["g1"] == 3 << 22)
        worst = max(worst, r["ndf_undecided"] / r["ndf"])
    assert worst < 0.01, worst          # the tier does decide (the wall-hugging family of sgd's g1 is undecided by design: not asserted)
    # ... and over EVERY float polar cosine of (0, 1] (bit patterns 1 .. 0x3f800000) for a few rows (tools/model_fast_exhaustive.py: all of them)
    for name in ("gold-metallic-paint", "green-acrylic", "alum-bronze", "white-marble"):
        b = getattr(djb, kind)(name, ctx=gpu_ctx)
        r = djb.selftest_model_fast(b, 0x3f800000, seed=0, first=1, ctx=gpu_ctx)
        assert r["g1_mismatch"] == 0 and r["ndf_mismatch"] == 0 and r["ndf"] == 3 * 0x3f800000, (name, r)


def test_sgd_fast_tier_equals_exact_chain(gpu_ctx, monkeypatch):
    """sgd::eval through the fast tier against the same kernel on objects created with the tier off (DJB_SGD_FAST=0): equal bits,
    also on a row outside the tier's domain (theta0 beyond 4: the flag is cleared at creation)."""
    n = 1 << 18
    i = synth.directions_aos(n, synth.SEED_I, 31); o = synth.directions_aos(n, synth.SEED_O, 31)
    for name in ("gold-metallic-paint", "green-acrylic", "ss440", "alumina-oxide"):
        fast = djb.sgd(name, ctx=gpu_ctx).eval(i, o)
        monkeypatch.setenv("DJB_SGD_FAST", "0")
        exact = djb.sgd(name, ctx=gpu_ctx).eval(i, o)
        monkeypatch.delenv("DJB_SGD_FAST")
        assert np.array_equal(fast.view(np.uint32), exact.view(np.uint32)), name
    row = np.array(param_tables.sgd_params("gold-metallic-paint"), np.float64)
    row[30:33] = (4.5, -4.5, 0.1)
    out = djb.sgd.from_params(row, ctx=gpu_ctx).eval(i, o)
    assert np.isfinite(out).all()
