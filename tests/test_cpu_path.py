"""The product's HOST execution path (dj_brdf_amd/csrc/djb_cpu.cpp: the kernels' per-unit code of djb_device.hpp
compiled for the CPU with the host libm), through the C ABI on a CPU context (djb_ctx_create(DJB_DEVICE_CPU)).

It exists for the reference's real callers -- scalar virtual calls from render threads, and machines without a GPU
(BASELINE.json configs[0]: "examples/merl_params.cpp ... CPU only, runs without a GPU") -- and it is the product's
own code: nothing under oracle/ is linked or imported by it.  Here it is checked, WITHOUT a GPU, against the CPU
oracle and the reference goldens: every value bit for bit (the host libm is the libm the reference calls, so on this
path even the fp64 trigonometry is identical by construction).  tests/test_gpu_scalar_path.py compares it with the GPU
batch path on the GPU box.
"""
import os
import subprocess

import numpy as np
import pytest

from dj_brdf_amd import djb, merl_params, synth
from golden_cases import ANISO_BIG_CASES, ANISO_CASES, FIT_CASES, PARAMS_TXT_MATERIALS, aniso_big_source
from test_gpu_parity import FRESNELS, PARAMS, assert_close, mk_fresnel, mk_params

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = 1 << 12


@pytest.fixture(scope="module")
def cpu():
    return djb.Context("cpu")


@pytest.fixture(scope="module")
def dirs():
    return (synth.directions_aos(N, synth.SEED_I), synth.directions_aos(N, synth.SEED_O),
            synth.uniforms(N, synth.SEED_U1), synth.uniforms(N, synth.SEED_U2))


def same(a, b):
    a, b = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
    return a.shape == b.shape and bool(((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))).all())


@pytest.mark.parametrize("ndf", ["ggx", "beckmann"])
@pytest.mark.parametrize("fres", FRESNELS, ids=lambda f: f[0])
def test_microfacet_operators(cpu, oracle, dirs, ndf, fres):
    i, o, u1, u2 = dirs
    g = getattr(djb, ndf)(mk_fresnel(fres), True, ctx=cpu)
    ob = oracle.microfacet(ndf, fres, True)
    for p in PARAMS:
        up = mk_params(p)
        for op in ("eval", "evalp", "pdf"):
            assert_close(f"{ndf}/{fres[0]}/{p}/{op}", getattr(g, op)(i, o, up), oracle.eval(ob, i, o, p, op))
        fr, pdf = g.eval_pdf(i, o, up)
        assert same(fr, oracle.eval(ob, i, o, p, "eval")) and same(pdf, oracle.eval(ob, i, o, p, "pdf"))
        assert same(g.sample(u1, u2, o, up), oracle.sample(ob, u1, u2, o, p)), (ndf, p, "sample")
        w, si, spdf = g.evalp_is(u1, u2, o, up)
        ww, wi, wpdf = oracle.evalp_is(ob, u1, u2, o, p)
        assert same(si, wi) and same(w, ww) and same(spdf, wpdf), (ndf, p, "evalp_is")


def test_small_and_strided_batches(cpu, oracle, dirs):
    """one pair at a time (what the facade's scalar virtuals send), ragged sizes, SoA and AoS"""
    i, o, _, _ = dirs
    g = djb.ggx(djb.fresnel.schlick((1.0, 0.71, 0.29)), True, ctx=cpu)
    p = djb.microfacet.params.isotropic(0.3)
    full = g.eval(i, o, p)
    for k in (0, 1, 63, 64, 65):
        assert same(g.eval(i[k:k + 1], o[k:k + 1], p), full[k:k + 1])
    assert same(g.eval(np.ascontiguousarray(i.T), np.ascontiguousarray(o.T), p).T, full)
    e = np.zeros((0, 3), np.float32)
    assert g.eval(e, e).shape == (0, 3)
    with pytest.raises(djb.exc):
        g.eval(i[:4], o[:5])


def test_merl_utia_lambert_models(cpu, oracle, dirs, tmp_path):
    i, o, u1, u2 = dirs
    tab = synth.merl_table_hashed()
    m, om = djb.merl.from_table(tab, ctx=cpu), oracle.merl_from_table(tab)
    assert np.array_equal(djb.merl_index(i, o, ctx=cpu), oracle.merl_index(i, o))
    for op in ("eval", "evalp", "pdf"):
        assert same(getattr(m, op)(i, o), oracle.eval(om, i, o, None, op)), op
    assert np.array_equal(m.get_samples(), np.ascontiguousarray(tab, np.float64).reshape(-1))
    p = str(tmp_path / "m.binary"); synth.write_merl_binary(p, tab)
    assert same(djb.merl(p, ctx=cpu).eval(i, o), m.eval(i, o))
    with pytest.raises(djb.exc) as e:
        djb.merl(str(tmp_path / "missing.binary"), ctx=cpu)
    assert e.value.status_name == "DJB_ERR_OPEN_FAILED" and "Failed to open" in str(e.value)
    raw = np.random.default_rng(11).uniform(-5.0, 120.0, size=3 * 288 * 288)
    up = str(tmp_path / "u.bin"); raw.tofile(up)
    u, ou = djb.utia.from_table(raw, ctx=cpu), oracle.utia(up)
    for op in ("eval", "evalp", "pdf"):
        assert same(getattr(u, op)(i, o), oracle.eval(ou, i, o, None, op)), op
    l, ol = djb.lambert(ctx=cpu), oracle.lambert()
    assert same(l.eval(i, o), oracle.eval(ol, i, o)) and same(l.sample(u1, u2, o), oracle.sample(ol, u1, u2, o))
    for kind in ("sgd", "abc"):
        b, ob = getattr(djb, kind)("gold-metallic-paint", ctx=cpu), getattr(oracle, kind)("gold-metallic-paint")
        assert same(b.eval(i, o), oracle.eval(ob, i, o)), kind
        # un-normalised directions: z > 1 makes sgd::g1's acos NaN, and djb::max(0.0, NaN) is NaN (not IEEE fmax's 0), so the
        # reference returns NaN; below-horizon and NaN inputs for completeness
        i3, o3 = (i * np.float32(3.0)).astype(np.float32), o.copy()
        o3[::7, 2] *= -1; i3[::11] = np.nan
        got, want = b.eval(i3, o3), oracle.eval(ob, i3, o3)
        assert same(got, want), kind
        if kind == "sgd":
            assert np.isnan(want).any() and not np.isnan(want).all()
    h, d = djb.brdf.io_to_hd(i, o, ctx=cpu)
    wh, wd = oracle.io_to_hd(i, o)
    assert same(h, wh) and same(d, wd)


@pytest.mark.parametrize("name", ["ggx90", "beckmann180", "merl_a30", "merl_a30_noshadow", "ggx7"])
def test_tabular_fit_golden(cpu, name):
    """djb::tabular(brdf, res, shadow) + both fits on the host against the REAL reference's tables (tests/golden/fit.npz)"""
    g = np.load(os.path.join(G, "fit.npz"))
    src, res, shadow = FIT_CASES[name]
    s = djb.merl.from_table(synth.merl_table(*src[1:]), ctx=cpu) if src[0] == "merl" else getattr(djb, src[0])(None, src[1], ctx=cpu)
    t = djb.tabular(s, res, shadow, ctx=cpu)
    for k, v in (("p22", t.get_p22v()), ("sigma", t.get_sigmav()), ("cdf", t.get_cdfv()), ("qf", t.get_qfv()),
                 ("fresnel", t.get_fresnel().get_points())):
        assert same(v, g[f"{name}_{k}"]), (name, k)
    ab = djb.tabular.fit_beckmann_parameters(t).get_ellipse()[0]
    ag = djb.tabular.fit_ggx_parameters(t).get_ellipse()[0]
    assert (np.float32(ab), np.float32(ag)) == (g[f"{name}_alpha_beckmann"][0], g[f"{name}_alpha_ggx"][0])
    assert same(t.eval(g["i"], g["o"]), g[f"{name}_eval"]) and same(t.pdf(g["i"], g["o"]), g[f"{name}_pdf"])
    assert same(t.sample(g["u1"], g["u2"], g["o"]), g[f"{name}_sample"])


@pytest.mark.parametrize("name", ["a_ggx", "a_merl"])
def test_anisotropic_fit_golden(cpu, name):
    from test_gpu_aniso import make_source
    g = np.load(os.path.join(G, "aniso.npz"))
    src, elev, azim, shadow = ANISO_CASES[name]
    t = djb.tabular_anisotropic(make_source(src, cpu), elev, azim, shadow, ctx=cpu)
    assert same(t.get_p22v()[0], g[f"{name}_p22"]) and same(t.get_sigmav()[0], g[f"{name}_sigma"])
    assert same(t.get_fresnel().get_points(), g[f"{name}_fresnel"])
    fb = np.array(djb.tabular_anisotropic.fit_beckmann_parameters(t).get_pdfparams(), np.float32)
    fg = np.array(djb.tabular_anisotropic.fit_ggx_parameters(t).get_pdfparams(), np.float32)
    assert same(fb, g[f"{name}_fit_beckmann"]) and same(fg, g[f"{name}_fit_ggx"])
    u1, u2 = g["u1"], g["u2"]
    phi, th = (u1 * np.float32(6.2)).astype(np.float32), (u2 * np.float32(1.5)).astype(np.float32)
    for q, args in (("pdf1", (phi,)), ("cdf1", (phi,)), ("qf1", (u1,)), ("pdf2", (th, phi)), ("cdf2", (th, phi)), ("qf2", (u2, phi))):
        assert same(getattr(t, q)(*args), g[f"{name}_{q}"]), (name, q)
    assert same(t.eval(g["i"], g["o"]), g[f"{name}_eval"]) and same(t.sample(u1, u2, g["o"]), g[f"{name}_sample"])


def test_anisotropic_short_rows_and_utia_source(cpu, oracle, tmp_path):
    g = np.load(os.path.join(G, "aniso_big.npz"))
    for name in ("a_short", "a_utia_small", "a90_utia"):      # the last: the reference's own 90 x 90 size (7.7 s and a 513 MB matrix there; 0.6 s here)
        src, elev, azim, shadow = ANISO_BIG_CASES[name]
        L = type("L", (), {"merl": type("M", (), {"from_table": staticmethod(lambda t: djb.merl.from_table(t, ctx=cpu))}),
                           "utia": type("U", (), {"from_table": staticmethod(lambda t: djb.utia.from_table(t, ctx=cpu))})})
        t = djb.tabular_anisotropic(aniso_big_source(L, src), elev, azim, shadow, ctx=cpu)
        assert same(t.get_p22v()[0], g[f"{name}_p22"]) and same(t.get_sigmav()[0], g[f"{name}_sigma"])
        want = oracle.aniso_sampling_tables(oracle.tabular_anisotropic(aniso_big_source(oracle, src, str(tmp_path)), elev, azim, shadow))
        for q in ("pdf1", "cdf1", "qf1", "pdf2", "cdf2", "qf2"):
            assert same(t.get_table(q), want[q]), (name, q)
        assert t.qf2_entries() == want["qf2_entries"]
        if name == "a_short":
            assert t.qf2_entries() < elev * azim
            assert same(t.qf2(g[f"{name}_qf2_u"], g[f"{name}_qf2_phi"]), g[f"{name}_qf2"])


@pytest.mark.parametrize("name", ["phong50", "phong3", "ward"])
def test_user_defined_brdf_is_fitted_from_host_samples(cpu, oracle, name):
    """the reference's extension point (dj_brdf.h:74-109) on the host path: tests/user_defined_cases.py"""
    import user_defined_cases
    user_defined_cases.check_user_defined_fits(cpu, oracle, name)
    user_defined_cases.check_sample_count_errors(cpu)
    if name == "ward":
        user_defined_cases.check_lambert_source(cpu, oracle)


def test_facade_user_classes_on_the_host_path():
    """The C++ facade's extension points without a GPU: oracle/ref_shim.cpp compiled against include/dj_brdf.h (oracle/_facade, the
    conformance harness of tests/test_gpu_facade_conformance.py) driven with DJB_DEVICE=cpu -- user-derived brdf / fresnel::impl
    classes against the oracle and the reference's golden, user-derived radial / microfacet NDF classes against the golden."""
    import sys
    shim = os.path.join(ROOT, "oracle", "_facade", "libdjb_facade_shim.so")
    if not os.path.exists(shim):
        pytest.skip("oracle/_facade/libdjb_facade_shim.so not built (make -C oracle facade)")
    code = (
        "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import oraclelib, test_gpu_facade_conformance as T\n"
        "from dj_brdf_amd import synth\n"
        "lib = oraclelib.CheckerLib(T.SHIM, 'ref_')\n"
        "inputs = (synth.directions_aos(T.N, synth.SEED_I, start=777), synth.directions_aos(T.N, synth.SEED_O, start=777),\n"
        "          synth.uniforms(T.N, synth.SEED_U1, start=777), synth.uniforms(T.N, synth.SEED_U2, start=777))\n"
        "T.test_user_defined_classes.__wrapped__(lib, oraclelib.oracle(), inputs) if hasattr(T.test_user_defined_classes, '__wrapped__') "
        "else T.test_user_defined_classes(lib, oraclelib.oracle(), inputs)\n"
        "T.test_user_defined_ndf_classes(lib)\n"
        "print('ok')\n" % (os.path.join(ROOT, "tests"), ROOT))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=dict(os.environ, DJB_DEVICE="cpu", DJB_QUIET="1"))
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout[-2000:] + r.stderr[-3000:]


def test_params_txt_on_the_cpu(cpu, tmp_path):
    """BASELINE configs[0]: the merl_params driver on a machine without a GPU -- byte-identical params.txt"""
    files = []
    for name, recipe in PARAMS_TXT_MATERIALS:
        p = str(tmp_path / (name + ".binary"))
        synth.write_merl_binary(p, synth.merl_table(*recipe)); files.append(p)
    ab, ag, timing = merl_params.fit_files_on(cpu, files)
    txt = merl_params.format_params_txt(files, list(zip(ab.tolist(), ag.tolist())))
    assert txt.encode() == open(os.path.join(G, "params_expected.txt"), "rb").read()
    assert 0 < timing["bytes"] < 3 * 6000 * 24          # only the entries the fit reads are fetched from the files
    # the Python CLI, as a user without a GPU would run it
    out = str(tmp_path / "params_cli.txt")
    assert merl_params.main(["--cpu", "-o", out] + files) == 0
    assert open(out, "rb").read() == open(os.path.join(G, "params_expected.txt"), "rb").read()
    # batch entry points agree with the one-object path
    mats = [djb.merl(f, ctx=cpu) for f in files]
    ab2, ag2 = djb.fit_brdf_batch(mats, 90, True, ctx=cpu)
    assert np.array_equal(ab2, ab) and np.array_equal(ag2, ag)
    with pytest.raises(djb.exc) as e:
        merl_params.fit_files_on(cpu, files[:1] + [str(tmp_path / "missing.binary")])
    assert e.value.status_name == "DJB_ERR_OPEN_FAILED"


def test_file_that_shrinks_under_the_gather(cpu, tmp_path, monkeypatch):
    """The file passes the size check, is mapped, and is then truncated by someone else: the reference, which fread()s,
    reports "Reading <file> failed" (dj_brdf.h:979-982); so does the gather, instead of dying on SIGBUS -- and the process
    goes on to fit the next file."""
    p = str(tmp_path / "shrinks.binary")
    synth.write_merl_binary(p, synth.merl_table(*PARAMS_TXT_MATERIALS[0][1]))
    observer = djb.set_file_map_observer(lambda path: os.truncate(path, 9000000))   # after the size check and the mapping
    try:
        with pytest.raises(djb.exc) as e:
            merl_params.fit_files_on(cpu, [p])
        assert e.value.status_name == "DJB_ERR_READ_FAILED" and str(e.value).strip() == f"djb_error: Reading {p} failed"
        assert os.path.getsize(p) == 9000000
    finally:
        djb.set_file_map_observer(None); del observer
    with pytest.raises(djb.exc) as e:                                    # now short before the mapping: same verdict
        merl_params.fit_files_on(cpu, [p])
    assert e.value.status_name == "DJB_ERR_READ_FAILED"
    synth.write_merl_binary(p, synth.merl_table(*PARAMS_TXT_MATERIALS[0][1]))
    ab, ag, _ = merl_params.fit_files_on(cpu, [p])
    assert np.isfinite(ab).all() and np.isfinite(ag).all()


def test_lean_and_queries(cpu, oracle, dirs):
    from golden_cases import LEAN_BASE, LEAN_CASES, lean_moments, lean_texels
    i, o, u1, u2 = dirs
    lean = lean_moments(N)
    b = djb.beckmann(djb.fresnel.schlick((1.0, 0.71, 0.29)), True, ctx=cpu)
    ob = oracle.microfacet("beckmann", ("schlick", 1.0, 0.71, 0.29), True)
    for scale, filtering, biased in LEAN_CASES:
        tex = lean_texels(lean, biased)
        got, gpp = djb._eval_lean(b, i, o, mk_params(LEAN_BASE), scale, tex, want="evalp", return_params=True,
                                  filtering=filtering, biased=biased)
        want, wpp = oracle.eval_lean(ob, i, o, LEAN_BASE, scale, tex, "evalp", filtering=filtering, biased=biased)
        assert same(gpp, wpp) and same(got, want), (scale, filtering, biased)
        gw, gi, gpdf, gpp = b.sample_lean(u1, u2, o, mk_params(LEAN_BASE), scale, tex, True, return_params=True,
                                          filtering=filtering, biased=biased)
        ww, wi, wpdf, wpp = oracle.sample_lean(ob, u1, u2, o, LEAN_BASE, scale, tex, True, filtering=filtering, biased=biased)
        assert same(gpp, wpp) and same(gi, wi) and same(gw, ww) and same(gpdf, wpdf), (scale, filtering, biased)
    # NaN moments stay NaN through lrep_to_params: djb::max(1e-5, NaN) = NaN and min(0.99, max(-0.99, NaN)) = NaN (dj_brdf.h:574-575)
    bad = lean.copy(); bad[0::5, 2] = np.nan; bad[1::5, 4] = np.nan; bad[2::5, 0] = np.inf
    got, gpp = djb._eval_lean(b, i, o, mk_params(LEAN_BASE), 0.7, bad, want="evalp", return_params=True)
    want, wpp = oracle.eval_lean(ob, i, o, LEAN_BASE, 0.7, bad, "evalp")
    assert same(gpp, wpp) and same(got, want) and np.isnan(wpp).any()
    pp = oracle.eval_lean(ob, i, o, LEAN_BASE, 0.7, lean, "pdf")[1]
    assert same(b.sample_pp(u1, u2, o, pp), oracle.sample_lean(ob, u1, u2, o, LEAN_BASE, 0.7, lean, False)[0])
    h = oracle.io_to_hd(i, o)[0]
    up, p = mk_params(PARAMS[2]), PARAMS[2]
    assert same(b.ndf(h, up), oracle.microfacet_query(ob, "ndf", h, params=p))
    assert same(b.gaf(h, i, o, up), oracle.microfacet_query(ob, "gaf", h, i, o, params=p))
    assert same(b.sigma(o, up), oracle.microfacet_query(ob, "sigma", o, params=p))
    u = np.clip(u1, 1e-4, 1 - 1e-4)
    assert same(b.qf_radial(u), oracle.radial_query(ob, "qf_radial", u))


def test_mixed_backends_are_rejected(cpu):
    if djb.device_count() == 0:
        with pytest.raises(djb.exc) as e:      # a GPU context is never silently replaced by the host path
            djb.Context(0)
        assert e.value.status_name == "DJB_ERR_NO_DEVICE"
    g = djb.ggx(ctx=cpu)
    assert g.ctx.is_cpu and djb._lib.load().djb_brdf_kind(g._h) == 1


def test_cpp_programs_run_without_a_gpu(tmp_path):
    """The C++ djb:: facade on the host path: examples/facade_check (known answers of SURVEY 8-N) and the
    merl_params driver with DJB_DEVICE=cpu; where /root/reference is mounted, the reference's OWN programs compiled
    unchanged against include/dj_brdf.h write what the reference binaries write."""
    env = dict(os.environ, DJB_DEVICE="cpu")
    exe = os.path.join(ROOT, "examples", "facade_check")
    if not os.path.exists(exe):
        pytest.skip("examples not built")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "MISMATCH" not in r.stdout, r.stdout + r.stderr
    files = []
    for name, recipe in PARAMS_TXT_MATERIALS:
        p = str(tmp_path / (name + ".binary"))
        synth.write_merl_binary(p, synth.merl_table(*recipe)); files.append(p)
    want = open(os.path.join(G, "params_expected.txt"), "rb").read()
    for mode in ([], ["-s"]):
        r = subprocess.run([os.path.join(ROOT, "examples", "merl_params")] + mode + files, cwd=str(tmp_path), capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stdout + r.stderr
        assert (tmp_path / "params.txt").read_bytes() == want, mode
        (tmp_path / "params.txt").unlink()
    # user-defined classes (examples/custom_brdf.cpp: BRDFs derived from djb::brdf, a Fresnel term derived from fresnel::impl,
    # written against the reference's interface): the bytes the REAL reference prints for the same source (make_reftests.sh)
    r = subprocess.run([os.path.join(ROOT, "examples", "custom_brdf")], capture_output=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr
    assert r.stdout == open(os.path.join(G, "reftests", "custom_brdf.txt"), "rb").read()
    rt = os.path.join(ROOT, "examples", "_reftests")
    if os.path.exists(os.path.join(rt, "plot_cdf")):
        for prog in ("plot_cdf", "plot_qf"):
            r = subprocess.run([os.path.join(rt, prog)], cwd=str(tmp_path), capture_output=True, text=True, timeout=600, env=env)
            assert r.returncode == 0, r.stdout + r.stderr
        for f in sorted(x for x in os.listdir(os.path.join(G, "reftests")) if x.startswith("eval_")):
            assert (tmp_path / f).read_bytes() == open(os.path.join(G, "reftests", f), "rb").read(), f
        r = subprocess.run([os.path.join(rt, "merl_params")] + files, cwd=str(tmp_path), capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0 and (tmp_path / "params.txt").read_bytes() == want


@pytest.mark.parametrize("ndf", ["ggx", "beckmann"])
def test_eval_divides_the_zero_vector_by_cos_i(cpu, oracle, dirs, ndf):
    """microfacet::eval = evalp / i.z (dj_brdf.h:1551-1555) even where evalp returned vec3(0): -0 for i below the horizon, NaN on it.
    The host path against the oracle on pairs with either direction below / on the horizon, signs of zeros included."""
    i, o, _, _ = dirs
    i, o = i[:4096].copy(), o[:4096].copy()
    i[:1024, 2] *= -1; o[1024:2048, 2] *= -1
    i[2048:2112, 2] = 0.0; i[2112:2176, 2] = -0.0; o[2176:2240, 2] = 0.0; i[2240:2304, 2] = np.nan
    def vb(a):
        a = np.ascontiguousarray(a, np.float32)
        return np.where(np.isnan(a), np.uint32(0x7fc00000), a.view(np.uint32))
    for shadow in (True, False):
        g = getattr(djb, ndf)(djb.fresnel.ideal(), shadow, ctx=cpu)
        ob = oracle.microfacet(ndf, ("ideal",), shadow)
        for p in (None, ("elliptic", 0.3, 0.3, 0.0), ("elliptic", 0.05, 0.05, 0.0)):
            for op in ("eval", "evalp", "pdf"):
                want = oracle.eval(ob, i, o, p, op)
                assert np.array_equal(vb(getattr(g, op)(i, o, mk_params(p))), vb(want)), (ndf, shadow, p, op)
            if shadow:
                w = oracle.eval(ob, i, o, p, "eval")
                assert np.all(np.signbit(w[:1024])) and np.all(np.isnan(w[2048:2176])), "the reference's own zeros are not what this test believes"


def test_every_kind_and_operator_on_hostile_pairs_host_path():
    """tools/hostile_parity_sweep.py --cpu: the same sweep through the product's host path"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "hostile_parity_sweep.py"), "--cpu"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert lines[-1] == "cases with a mismatch: 0" and sum(l.endswith(" ok") for l in lines) >= 240, "\n".join(l for l in lines if not l.endswith(" ok"))


BOUNCE = r"""
import faulthandler, mmap, os, sys
sys.path.insert(0, {root!r})
from dj_brdf_amd import djb, synth, merl_params
cpu = djb.cpu_context()
p = {path!r}
synth.write_merl_binary(p, synth.merl_table(0.3))
merl_params.fit_files_on(cpu, [p])            # installs the library's SIGBUS guard
faulthandler.enable()                         # a handler installed AFTER ours; when it fires it restores its predecessor (ours) and re-raises
merl_params.fit_files_on(cpu, [p])            # the guard is re-installed on top, chaining to faulthandler's
# a SIGBUS that is none of the library's business: touch a mapped page beyond the end of a truncated file
q = p + ".other"
open(q, "wb").write(b"x" * 65536)
f = open(q, "r+b"); m = mmap.mmap(f.fileno(), 65536); os.truncate(q, 0)
print("touching", flush=True)
m[40000]                                      # SIGBUS here: the process must die of it, not spin between two handlers
print("survived", flush=True)
"""


def test_foreign_sigbus_does_not_bounce_between_handlers(tmp_path):
    """a handler installed after the library's guard that falls back to the guard (faulthandler restores its predecessor and
    re-raises) used to ping-pong with on_sigbus forever on a SIGBUS outside the library's mappings.  The process must terminate with
    SIGBUS."""
    import signal, subprocess, sys
    src = BOUNCE.format(root=ROOT, path=str(tmp_path / "m.binary"))
    try:
        r = subprocess.run([sys.executable, "-c", src], capture_output=True, text=True, timeout=120,
                           env=dict(os.environ, DJB_QUIET="1"))
    except subprocess.TimeoutExpired:
        pytest.fail("the process hung on a foreign SIGBUS (handlers bouncing)")
    assert "touching" in r.stdout and "survived" not in r.stdout
    assert r.returncode == -signal.SIGBUS, (r.returncode, r.stderr[-500:])


def test_on_chip_generator_restatement_and_quality(cpu):
    """The round-6 on-chip uniforms (csrc/djb_device_units.inc: gen_uniform; synth.rng_uniforms restates it): the host path's bits equal
    the numpy restatement, and the stream behaves like uniforms where the sampler can tell -- 2-D equidistribution of the (u1, u2) pairs
    the Beckmann bench draws (64 x 64 cells, SURVEY 8d's histogram), serial correlation, per-bit balance, distinct streams per seed."""
    n = 1 << 20
    for seed, start in ((synth.SEED_U1, 0), (synth.SEED_U2, (1 << 33) + 7), (0, 123456789)):
        got = djb.gen_uniforms(n, seed, start=start, ctx=cpu)
        assert np.array_equal(np.asarray(got, np.float32).view(np.uint32), synth.rng_uniforms(n, seed, start=start).view(np.uint32))
    n = 1 << 22
    u1, u2 = synth.rng_uniforms(n, synth.SEED_U1), synth.rng_uniforms(n, synth.SEED_U2)
    assert 0.0 <= u1.min() and u1.max() < 1.0
    # 2-D chi-square, 4095 degrees of freedom: mean 4095, sigma 90.5
    h = np.histogram2d(u1, u2, bins=64, range=((0, 1), (0, 1)))[0]
    e = n / 4096.0
    chi2 = float(((h - e) ** 2 / e).sum())
    assert abs(chi2 - 4095.0) < 5 * 90.5, chi2
    # pairs of CONSECUTIVE counters of one stream (what a stratified caller would see)
    h = np.histogram2d(u1[:-1], u1[1:], bins=64, range=((0, 1), (0, 1)))[0]
    chi2 = float(((h - (n - 1) / 4096.0) ** 2 / ((n - 1) / 4096.0)).sum())
    assert abs(chi2 - 4095.0) < 5 * 90.5, chi2
    h = np.histogram2d(u1[:-64], u1[64:], bins=64, range=((0, 1), (0, 1)))[0]          # the same lane of consecutive waves
    chi2 = float(((h - (n - 64) / 4096.0) ** 2 / ((n - 64) / 4096.0)).sum())
    assert abs(chi2 - 4095.0) < 5 * 90.5, chi2
    for a, b in ((u1[:-1], u1[1:]), (u1, u2), (u1[:-64], u1[64:])):       # serial / cross / lane-stride correlation: sigma = 1/sqrt(n)
        r = float(np.corrcoef(a.astype(np.float64), b.astype(np.float64))[0, 1])
        assert abs(r) < 5.0 / np.sqrt(n), r
    bits = (u1 * np.float32(2.0 ** 24)).astype(np.uint32)
    for b in range(24):                                                    # every one of the 24 bits is fair: sigma = 0.5/sqrt(n)
        f = float(((bits >> np.uint32(b)) & np.uint32(1)).mean())
        assert abs(f - 0.5) < 5 * 0.5 / np.sqrt(n), (b, f)
    assert not np.array_equal(u1, synth.rng_uniforms(n, synth.SEED_U1 + 1))


def test_fit_files_over_several_contexts(tmp_path):
    """djb_fit_merl_files_multi (ABI 233): file k on context k mod G, one host thread per context inside the library,
    rows in input order, the lowest-indexed bad file's error wins, a context listed twice is refused.  CPU contexts here; the GPU
    form of the same test (2 and 3 contexts on one device) is tests/test_gpu_golden.py::test_fit_files_multi_contexts."""
    recipes = [synth.material_recipe(k) for k in range(4)]
    paths = []
    for k in range(7):
        p = str(tmp_path / f"m{k}.binary")
        synth.write_merl_binary(p, synth.merl_table(*recipes[k % 4])); paths.append(p)
    one = djb.Context("cpu")
    ab1, ag1, _ = merl_params.fit_files_on(one, paths)
    for g in (1, 2, 3, 9):                              # 9 > number of files: the spare contexts stay idle
        ctxs = [djb.Context("cpu") for _ in range(g)]
        ab, ag, per = merl_params.fit_files_multi(ctxs, paths)
        assert np.array_equal(ab.view(np.uint32), ab1.view(np.uint32)) and np.array_equal(ag.view(np.uint32), ag1.view(np.uint32)), g
        assert len(per) == g and sum(t["bytes"] for t in per) == 7 * 5545 * 24
    # errors: files 1 (context 1: bad header) and 2 (context 0: missing) are bad -> the reference's loop would stop at file 1
    bad = tmp_path / "bad.binary"; bad.write_bytes(np.array([90, 90, 181], np.int32).tobytes() + b"\0" * 100)
    broken = [paths[0], str(bad), str(tmp_path / "missing.binary")] + paths[3:]
    ctxs = [djb.Context("cpu"), djb.Context("cpu")]
    with pytest.raises(djb.exc) as e:
        merl_params.fit_files_multi(ctxs, broken)
    assert e.value.status_name == "DJB_ERR_BAD_HEADER", str(e.value)
    with pytest.raises(djb.exc) as e:
        merl_params.fit_files_multi(ctxs, [paths[0], paths[1], str(tmp_path / "missing.binary"), str(bad)])
    assert e.value.status_name == "DJB_ERR_OPEN_FAILED" and "missing.binary" in str(e.value)
    with pytest.raises(djb.exc) as e:
        merl_params.fit_files_multi([ctxs[0], ctxs[0]], paths)
    assert e.value.status_name == "DJB_ERR_INVALID_ARGUMENT"
    assert merl_params.fit_files_multi(ctxs, [])[0].size == 0
    # the driver's own entry point goes through the same call
    out = merl_params.fit_files(paths, cpu=True)
    assert [a for a, _ in out] == [float(a) for a in ab1]
