"""djb::tabular_anisotropic on the HIP path against golden vectors from the
real reference: fit tables, two-level sampling tables (through the public queries), the two
5-parameter moment fits, and the operators of the fitted object."""
import os

import numpy as np
import pytest

from dj_brdf_amd import djb, synth
from golden_cases import ANISO_CASES
from test_gpu_parity import assert_close

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def make_source(src, ctx):
    if src[0] == "abc":
        return djb.abc(src[1], ctx=ctx)
    if src[0] == "merl":
        return djb.merl.from_table(synth.merl_table(*src[1:]), ctx=ctx)
    return getattr(djb, src[0])(None, src[1], ctx=ctx)


@pytest.mark.parametrize("name", list(ANISO_CASES))
def test_anisotropic_fit_golden(gpu_ctx, name):
    g = np.load(os.path.join(G, "aniso.npz"))
    src, elev, azim, shadow = ANISO_CASES[name]
    t = djb.tabular_anisotropic(make_source(src, gpu_ctx), elev, azim, shadow, ctx=gpu_ctx)
    p22, e, a = t.get_p22v()
    assert (e, a) == (elev, azim)
    assert_close(f"{name}/p22", p22, g[f"{name}_p22"], 2e-5)
    assert_close(f"{name}/sigma", t.get_sigmav()[0], g[f"{name}_sigma"], 2e-5)
    assert_close(f"{name}/fresnel", t.get_fresnel().get_points(), g[f"{name}_fresnel"], 2e-5)
    fb = np.array(djb.tabular_anisotropic.fit_beckmann_parameters(t).get_pdfparams(), np.float32)
    fg = np.array(djb.tabular_anisotropic.fit_ggx_parameters(t).get_pdfparams(), np.float32)
    # rho / mux / muy are ~1e-7 residues of cancelling sums: only the reference's summation order gives them
    assert_close(f"{name}/fit_beckmann", fb, g[f"{name}_fit_beckmann"])
    assert_close(f"{name}/fit_ggx", fg, g[f"{name}_fit_ggx"])
    u1, u2 = g["u1"], g["u2"]
    phi, th = (u1 * np.float32(6.2)).astype(np.float32), (u2 * np.float32(1.5)).astype(np.float32)
    for q, args in (("pdf1", (phi,)), ("cdf1", (phi,)), ("qf1", (u1,)), ("pdf2", (th, phi)),
                    ("cdf2", (th, phi)), ("qf2", (u2, phi))):
        assert_close(f"{name}/{q}", getattr(t, q)(*args), g[f"{name}_{q}"], 5e-5)
    for op in ("eval", "evalp", "pdf"):
        assert_close(f"{name}/{op}", getattr(t, op)(g["i"], g["o"]), g[f"{name}_{op}"], 1e-4)
    assert_close(f"{name}/eval elliptic", t.eval(g["i"], g["o"], djb.microfacet.params.elliptic(0.2, 0.5, 0.7)),
                 g[f"{name}_eval_ell"], 1e-4)
    assert_close(f"{name}/sample", t.sample(u1, u2, g["o"]), g[f"{name}_sample"])


def test_anisotropic_full_resolution_vs_oracle(gpu_ctx, oracle):
    """The reference's own size class: 90 x 90 would build an 8010^2 (513 MB) matrix on the CPU;
    a 40 x 48 fit keeps the oracle to seconds while exercising the same code (N = 1872)."""
    elev, azim = 40, 48
    t = djb.tabular_anisotropic(djb.ggx(ctx=gpu_ctx), elev, azim, True, ctx=gpu_ctx)
    ot = oracle.tabular_anisotropic(oracle.microfacet("ggx"), elev, azim, True)
    want = oracle.aniso_tables(ot)
    assert_close("p22", t.get_p22v()[0], want["p22"], 2e-5)
    assert_close("sigma", t.get_sigmav()[0], want["sigma"], 2e-5)
    fb = np.array(djb.tabular_anisotropic.fit_beckmann_parameters(t).get_pdfparams(), np.float32)
    assert_close("fit_beckmann", fb, np.asarray(want["fit_beckmann"], np.float32))


@pytest.mark.parametrize("name", ["a90_merl", "a90_utia", "a_utia_small", "a_short", "a_short12"])
def test_anisotropic_big_utia_and_short_rows(gpu_ctx, oracle, name, tmp_path):
    """The reference's own 90 x 90 size (8010^2-double matrix on the CPU), UTIA-sourced fits, and fits whose
    conditional quantile table comes up short (dj_brdf.h:3005-3034): the HIP path against goldens from the REAL
    reference (tests/golden/aniso_big.npz) and, table by table, against the oracle -- including the reference's
    shifted m_qf2 layout (default) and the aligned layout behind DJB_OPT_ANISO_QF2_ALIGNED."""
    from golden_cases import ANISO_BIG_CASES, aniso_big_source
    g = np.load(os.path.join(G, "aniso_big.npz"))
    src, elev, azim, shadow = ANISO_BIG_CASES[name]
    dsrc = aniso_big_source(djb, src)
    t = djb.tabular_anisotropic(dsrc, elev, azim, shadow, ctx=gpu_ctx)
    p22, e, a = t.get_p22v()
    assert (e, a) == (elev, azim)
    assert_close(f"{name}/p22", p22, g[f"{name}_p22"], 2e-5)
    assert_close(f"{name}/sigma", t.get_sigmav()[0], g[f"{name}_sigma"], 2e-5)
    assert_close(f"{name}/fresnel", t.get_fresnel().get_points(), g[f"{name}_fresnel"], 2e-5)
    fb = np.array(djb.tabular_anisotropic.fit_beckmann_parameters(t).get_pdfparams(), np.float32)
    fg = np.array(djb.tabular_anisotropic.fit_ggx_parameters(t).get_pdfparams(), np.float32)
    assert_close(f"{name}/fit_beckmann", fb, g[f"{name}_fit_beckmann"])
    assert_close(f"{name}/fit_ggx", fg, g[f"{name}_fit_ggx"])
    u1, u2 = g["u1"], g["u2"]
    phi, th = (u1 * np.float32(6.2)).astype(np.float32), (u2 * np.float32(1.5)).astype(np.float32)
    for q, args in (("pdf1", (phi,)), ("cdf1", (phi,)), ("qf1", (u1,)), ("pdf2", (th, phi)), ("cdf2", (th, phi))):
        assert_close(f"{name}/{q}", getattr(t, q)(*args), g[f"{name}_{q}"], 5e-5)
    for op in ("eval", "pdf"):
        assert_close(f"{name}/{op}", getattr(t, op)(g["i"], g["o"]), g[f"{name}_{op}"], 1e-4)
    short = name.startswith("a_short")
    if short:
        entries = int(g[f"{name}_qf2_entries"][0])
        assert t.qf2_entries() == entries < elev * azim
        assert_close(f"{name}/qf2 (taps the reference holds)", t.qf2(g[f"{name}_qf2_u"], g[f"{name}_qf2_phi"]), g[f"{name}_qf2"], 5e-5)
    else:
        assert t.qf2_entries() == elev * azim
        assert_close(f"{name}/qf2", t.qf2(u2, phi), g[f"{name}_qf2"], 5e-5)
        assert_close(f"{name}/sample", t.sample(u1, u2, g["o"]), g[f"{name}_sample"])
    # every sampling table against the restatement (which the CPU suite pins to the same goldens)
    ot = oracle.tabular_anisotropic(aniso_big_source(oracle, src, str(tmp_path)), elev, azim, shadow)
    want = oracle.aniso_sampling_tables(ot)
    for q in ("pdf1", "cdf1", "qf1", "pdf2", "cdf2", "qf2"):
        assert_close(f"{name}/table {q}", t.get_table(q), want[q])
    if short:
        # the aligned layout: same rows, each at its own offset, padded with 1.0
        djb.set_aniso_qf2_aligned(gpu_ctx, True)
        try:
            ta = djb.tabular_anisotropic(dsrc, elev, azim, shadow, ctx=gpu_ctx)
        finally:
            djb.set_aniso_qf2_aligned(gpu_ctx, False)
        qa, qr = ta.get_table("qf2").reshape(azim, elev), t.get_table("qf2")
        off = 0
        for k in range(azim):
            n_k = int(np.argmax(qa[k] == 1.0)) + 1            # entries up to and including the closing 1.0
            assert np.array_equal(qa[k, :n_k], qr[off:off + n_k]) and (qa[k, n_k:] == 1.0).all(), k
            off += n_k
        assert off == entries and (qr[off:] == 1.0).all()
        for q in ("pdf2", "cdf2", "qf1"):
            assert np.array_equal(ta.get_table(q), t.get_table(q))


def test_anisotropic_queries_reject_other_kinds(gpu_ctx):
    g = djb.ggx(ctx=gpu_ctx)
    with pytest.raises(djb.exc):
        djb.tabular_anisotropic.pdf1(g, np.zeros(4, np.float32))
    with pytest.raises(djb.exc):
        djb.tabular_anisotropic(g, 1, 8, ctx=gpu_ctx)              # "Invalid Resolution", dj_brdf.h:2244


def test_anisotropic_grids_beyond_the_lds_budget(gpu_ctx):
    """k_eval / k_sample stage a fitted lobe's tables in LDS when they fit (two 90 x 90 grids do); at 136 x 128 sigma's grid fits and
    the slope-pdf grid does not, at 160 x 128 neither does: every combination must give what the per-unit code gives without any staging --
    the host twin that answers calls of <= 96 units (same object, tables read back from HBM)."""
    n = 1 << 15
    i, o = synth.directions_aos(n, synth.SEED_I), synth.directions_aos(n, synth.SEED_O)
    u1, u2 = synth.uniforms(n, synth.SEED_U1), synth.uniforms(n, synth.SEED_U2)
    for elev, azim in ((136, 128), (160, 128)):
        t = djb.tabular_anisotropic(djb.ggx(ctx=gpu_ctx), elev, azim, True, ctx=gpu_ctx)
        fr, pdf = t.eval_pdf(i, o)                                   # batch: the kernels
        s = t.sample(u1, u2, o)
        assert np.isfinite(fr).all() and float(np.abs(fr).sum()) > 0
        for lo in (0, 7777, n - 64):                                 # 64 units per call: answered by the host twin
            sl = slice(lo, lo + 64)
            fr_h, pdf_h = t.eval_pdf(i[sl], o[sl])
            assert np.array_equal(fr[sl].view(np.uint32), fr_h.view(np.uint32)), (elev, azim, lo)
            assert np.array_equal(pdf[sl].view(np.uint32), pdf_h.view(np.uint32)), (elev, azim, lo)
            assert np.array_equal(s[sl].view(np.uint32), t.sample(u1[sl], u2[sl], o[sl]).view(np.uint32)), (elev, azim, lo)
