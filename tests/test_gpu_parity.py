"""HIP path vs the CPU oracle, through the C ABI (ctypes -> libdjb_hip.so), on seeded inputs.

Bar (BASELINE.json north_star): float BRDF values within 1e-5 relative of the reference CPU
path; MERL bin indices bit-exact.  The oracle is bit-exact with the real reference
(tests/test_oracle_vs_ref.py, tests/test_oracle_golden.py), so oracle parity == reference parity.
"""
import os

import numpy as np
import pytest

from dj_brdf_amd import djb, synth

pytestmark = pytest.mark.gpu

RTOL = 1e-5          # north_star tolerance
ATOL = 1e-30         # values that underflow to (sub)denormal noise


def rel_err(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.abs(a - b) / np.maximum(np.abs(b), ATOL / RTOL)


def assert_close(name, got, want, rtol=RTOL):
    """The contract is `rtol` relative (north_star: 1e-5; looser figures at call sites date from before the
    kernels matched the reference's arithmetic operation for operation).  Measured on MI355X: every comparison
    in this suite is bit-identical, so THAT is what is asserted; `rtol` only classifies the failure message."""
    got = np.asarray(got); want = np.asarray(want)
    assert got.shape == want.shape, f"{name}: shape {got.shape} vs {want.shape}"
    assert np.array_equal(np.isnan(got), np.isnan(want)), f"{name}: NaN mismatch"
    m = ~np.isnan(want)
    same = got.view(np.uint32)[m] == want.view(np.uint32)[m]
    if not same.all():
        e = rel_err(got[m], want[m])
        raise AssertionError(f"{name}: {np.mean(~same):.3e} of values not bit-identical to the reference's; max rel err "
                             f"{e.max():.3e} ({'within' if e.max() <= rtol else 'OUTSIDE'} the {rtol} contract)")
    return 1.0


N = 1 << 17
FRESNELS = [
    ("ideal",), ("unpolarized", 1.5, 1.8, 2.4), ("schlick", 1.0, 0.71, 0.29),
    ("sgd", 0.8, 0.5, 0.3, 0.1, 0.05, 0.02),
    ("spline",) + tuple(np.linspace(0.2, 1.0, 30, dtype=np.float32).repeat(3).tolist()),
]
PARAMS = [None, ("elliptic", 0.3, 0.3, 0.0), ("elliptic", 0.2, 0.5, 0.7),
          ("pdfparams", 0.4, 0.25, 0.3, 0.1, -0.05)]


def mk_fresnel(f):
    k = f[0]
    if k == "ideal": return djb.fresnel.ideal()
    if k == "unpolarized": return djb.fresnel.unpolarized(f[1:4])
    if k == "schlick": return djb.fresnel.schlick(f[1:4])
    if k == "sgd": return djb.fresnel.sgd(f[1:4], f[4:7])
    return djb.fresnel.spline(np.array(f[1:], np.float32).reshape(-1, 3))


def mk_params(p):
    if p is None: return None
    if p[0] == "elliptic": return djb.microfacet.params.elliptic(*p[1:])
    return djb.microfacet.params.pdfparams(*p[1:])


@pytest.fixture(scope="module")
def dirs():
    i = synth.directions_aos(N, synth.SEED_I); o = synth.directions_aos(N, synth.SEED_O)
    u1 = synth.uniforms(N, synth.SEED_U1); u2 = synth.uniforms(N, synth.SEED_U2)
    return i, o, u1, u2


@pytest.mark.parametrize("ndf", ["ggx", "beckmann"])
@pytest.mark.parametrize("fres", FRESNELS, ids=lambda f: f[0])
def test_microfacet_eval_pdf(gpu_ctx, oracle, dirs, ndf, fres):
    i, o, _, _ = dirs
    for shadow in ((True, False) if fres[0] == "ideal" else (True,)):
        g = getattr(djb, ndf)(mk_fresnel(fres), shadow, ctx=gpu_ctx)
        ob = oracle.microfacet(ndf, fres, shadow)
        for p in PARAMS:
            up = mk_params(p)
            for op in ("eval", "evalp", "pdf"):
                got = getattr(g, op)(i, o, up)
                ex = assert_close(f"{ndf}/{fres[0]}/{shadow}/{p}/{op}", got, oracle.eval(ob, i, o, p, op))
                # stronger than the 1e-5 contract: the kernels keep the reference's float/double
                # evaluation order, so outputs are bit-identical
                assert ex == 1.0
            fr, pdf = g.eval_pdf(i, o, up)
            assert_close("fused eval", fr, oracle.eval(ob, i, o, p, "eval"))
            assert_close("fused pdf", pdf, oracle.eval(ob, i, o, p, "pdf"))


@pytest.mark.parametrize("ndf", ["ggx", "beckmann"])
def test_microfacet_sample(gpu_ctx, oracle, dirs, ndf):
    _, o, u1, u2 = dirs
    g = getattr(djb, ndf)(djb.fresnel.schlick((1.0, 0.71, 0.29)), True, ctx=gpu_ctx)
    ob = oracle.microfacet(ndf, ("schlick", 1.0, 0.71, 0.29), True)
    for p in PARAMS:
        up = mk_params(p)
        # the Newton inversion stops at |value| < 1e-5 (dj_brdf.h:1938), so one differing ulp in a float
        # transcendental can move a sample by ~1e-4.  The kernels therefore run glibc's own logf / expf /
        # powf algorithms (djb_device.hpp glibc_*; pinned to the host libm by
        # test_oracle_golden.py::test_glibc_float_libm_restatement): every sample is bit-identical.
        got = g.sample(u1, u2, o, up)
        want = oracle.sample(ob, u1, u2, o, p)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), \
            f"{ndf} sample {p}: {np.mean(got.view(np.uint32) != want.view(np.uint32)):.2e} of components differ"
        w, gi, pdf = g.evalp_is(u1, u2, o, up)
        ww, wi, wpdf = oracle.evalp_is(ob, u1, u2, o, p)
        assert np.array_equal(gi.view(np.uint32), wi.view(np.uint32)), f"{ndf} evalp_is direction {p}"
        for name, a, b in (("weight", w, ww), ("pdf", pdf, wpdf)):
            same = (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))
            assert same.all(), f"{ndf} evalp_is {name} {p}: {np.mean(~same):.2e} differ"


def _beckmann_sampler_cases(n_bulk=150_001):
    """Inputs that leave the common path of the Beckmann sampling kernel (djb_kernels_sample.hip) for every reason it
    knows -- erfinv's tail arm at both call sites, more than four Newton trips, the clamp of the final argument,
    exp(-cot^2) below glibc's main path, directions on / below / along the horizon and on the normal, NaNs -- mixed
    into a random bulk so that dense waves, ragged tails and partly filled deferred queues all occur."""
    rng = np.random.default_rng(77)
    o = synth.directions_aos(n_bulk, synth.SEED_O).copy()
    u1 = synth.uniforms(n_bulk, synth.SEED_U1).copy(); u2 = synth.uniforms(n_bulk, synth.SEED_U2).copy()
    k = rng.permutation(n_bulk)
    def take(m):
        nonlocal k
        sel, k = k[:m], k[m:]
        return sel
    edge = np.array([0.0, 1e-7, 1e-6, 1e-5, 1e-3, 0.5, 1 - 1e-3, 1 - 1e-5, 1 - 1e-6, 1 - 6e-8, 1.0], np.float32)
    u2[take(4000)] = rng.choice(edge, 4000)                                   # qf1: tail arm, |2u - 1| -> 1
    u2[take(4000)] = np.float32(1) - rng.random(4000, dtype=np.float32) * np.float32(4e-3)
    u1[take(4000)] = rng.choice(edge, 4000)                                   # qf2: first guess at the ends of [-1, c]
    u1[take(4000)] = rng.random(4000, dtype=np.float32) * np.float32(1e-4)    # many trips / bisection steps
    u1[take(4000)] = np.float32(1) - rng.random(4000, dtype=np.float32) * np.float32(1e-4)
    s = take(3000); o[s] = (0, 0, 1)                                          # on the normal: sin_k = 0
    s = take(3000); t = rng.random(3000) * 1e-3; ph = rng.random(3000) * 6.2831853
    o[s] = np.stack([np.sin(t) * np.cos(ph), np.sin(t) * np.sin(ph), np.cos(t)], 1).astype(np.float32)   # cot^2 >= 512 and beyond 745
    s = take(3000); z = (rng.random(3000) * 2e-3).astype(np.float32); ph = rng.random(3000) * 6.2831853
    o[s] = np.stack([np.sqrt(1 - z * z) * np.cos(ph), np.sqrt(1 - z * z) * np.sin(ph), z], 1).astype(np.float32)   # grazing
    s = take(1000); o[s, 2] = -np.abs(o[s, 2])                                # below the horizon -> (0, 0, 1)
    s = take(200); o[s, 0] = np.nan
    s = take(200); u1[s] = np.nan
    s = take(200); u2[s] = np.nan
    s = take(200); o[s] = 0.0
    return o, u1, u2


def _same_bits(a, b):
    a = np.asarray(a, np.float32); b = np.asarray(b, np.float32)
    return (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))


BK_SAMPLER_PARAMS = PARAMS + [("elliptic", 0.01, 0.01, 0.0), ("elliptic", 1.0, 1.0, 0.0), ("elliptic", 0.05, 0.8, 0.9)]


@pytest.mark.parametrize("p", BK_SAMPLER_PARAMS, ids=lambda p: "default" if p is None else "-".join(str(x) for x in p))
def test_beckmann_sampler_deferred_paths(gpu_ctx, oracle, p):
    """sample / evalp_is of the Beckmann lobe on inputs chosen to leave the kernel's common path: every bit equals the oracle's,
    for device-resident dense ([3, n]) and strided ([n, 3]) batches and for host batches, including batch sizes that end in
    a partial wave."""
    import torch
    o, u1, u2 = _beckmann_sampler_cases()
    g = djb.beckmann(djb.fresnel.schlick((1.0, 0.71, 0.29)), True, ctx=gpu_ctx)
    ob = oracle.microfacet("beckmann", ("schlick", 1.0, 0.71, 0.29), True)
    up = mk_params(p)
    want = oracle.sample(ob, u1, u2, o, p)
    ww, wi, wpdf = oracle.evalp_is(ob, u1, u2, o, p)
    dev = torch.device("cuda", 0)
    tu1, tu2 = torch.as_tensor(u1, device=dev), torch.as_tensor(u2, device=dev)
    for layout in ("host", "dense", "strided"):
        oo = o if layout == "host" else torch.as_tensor(o, device=dev) if layout == "strided" else torch.as_tensor(np.ascontiguousarray(o.T), device=dev)
        a1, a2 = (u1, u2) if layout == "host" else (tu1, tu2)
        fix = (lambda x: x) if layout == "host" else (lambda x: x.cpu().numpy()) if layout == "strided" else (lambda x: x.cpu().numpy().T if x.dim() == 2 else x.cpu().numpy())
        got = fix(g.sample(a1, a2, oo, up))
        bad = ~_same_bits(got, want).all(axis=1)
        assert not bad.any(), f"sample {layout} {p}: {int(bad.sum())} differ, first {np.flatnonzero(bad)[:5]} o={o[bad][:2]} u1={u1[bad][:2]} u2={u2[bad][:2]}"
        w, gi, pdf = g.evalp_is(a1, a2, oo, up)
        w, gi, pdf = fix(w), fix(gi), fix(pdf)
        assert _same_bits(gi, wi).all(), f"evalp_is direction {layout} {p}"
        assert _same_bits(w, ww).all(), f"evalp_is weight {layout} {p}"
        assert _same_bits(pdf, wpdf).all(), f"evalp_is pdf {layout} {p}"
    # on-chip uniforms (sample_rng), dense and strided device batches: the same bits as the array path fed with gen_uniforms
    n = o.shape[0]
    g1, g2 = djb.gen_uniforms(n, synth.SEED_U1, ctx=gpu_ctx), djb.gen_uniforms(n, synth.SEED_U2, ctx=gpu_ctx)
    od, os_ = torch.as_tensor(np.ascontiguousarray(o.T), device=dev), torch.as_tensor(o, device=dev)
    ref = g.sample(g1, g2, od, up)
    assert torch.equal(g.sample_rng(synth.SEED_U1, synth.SEED_U2, od, up).view(torch.int32), ref.view(torch.int32)), f"sample_rng dense {p}"
    assert torch.equal(g.sample_rng(synth.SEED_U1, synth.SEED_U2, os_, up).T.contiguous().view(torch.int32), ref.view(torch.int32)), f"sample_rng strided {p}"
    # short device batches: one partial wave, one wave + 1, a workgroup + 1
    for m in (1, 63, 65, 97, 257, 1025):
        oo = torch.as_tensor(np.ascontiguousarray(o[-m:].T), device=dev)
        got = g.sample(tu1[-m:].contiguous(), tu2[-m:].contiguous(), oo, up).cpu().numpy().T
        assert _same_bits(got, want[-m:]).all(), f"sample, {m} units, {p}"


def test_device_libm_restatements(gpu_ctx, oracle):
    """The kernels' own copies of glibc's exp / pow / atan2 / sin / cos / tan / acos (double) and logf / expf / powf (float), evaluated on the GPU
    (djb_selftest_libm), against the libm of this host -- what the reference calls.  Every bit."""
    from test_oracle_golden import libm_f64_cases
    for fn, sets in libm_f64_cases(n=1 << 19).items():
        for x, y in sets:
            want = oracle.libm_f64(fn, x, y)
            got = djb.selftest_libm(("exp", "pow", "atan2", "sin", "cos", "tan", "acos")[fn], x, y, ctx=gpu_ctx)
            same = (want.view(np.uint64) == got.view(np.uint64)) | (np.isnan(want) & np.isnan(got))
            if fn == 1:      # pow: negative and subnormal bases are left to the device libm (never reached by the BRDF code): 1 ulp
                other = ((np.abs(x) < 2.3e-308) & (x != 0)) | (x < 0)
                with np.errstate(all="ignore"):
                    close = np.abs(got - want) <= 4 * np.spacing(np.abs(want))
                same |= other & (close | (np.isinf(want) & (want == got)))
            if 3 <= fn <= 5:      # sin / cos beyond 105414350 (__branred) and tan beyond 25 are left to the device libm (the BRDF code's angles stay below 7): 1 ulp
                with np.errstate(all="ignore"):
                    same |= (np.abs(x) >= (25.0 if fn == 5 else 105414350.0)) & (np.abs(got - want) <= 2 * np.spacing(np.abs(want)))
            assert same.all(), (fn, int((~same).sum()), x[~same][:3], y[~same][:3] if y is not None else None)
    # the kernels' float(atan2(y, x)) and float(r2d * atan2(y, x)) of float arguments (device libm first, glibc's algorithm
    # next to a float rounding boundary: djb_device.hpp atan2_to_f32) against the same expressions with the host libm
    rng = np.random.default_rng(5)
    n = 1 << 21
    f32 = lambda a: np.asarray(a, np.float32)
    anyf = lambda: rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32).view(np.float32)
    xx = f32(rng.uniform(-4, 4, n))
    r2d = np.float64(np.float32(180.0 / np.pi))
    with np.errstate(all="ignore"):
        fams = [(f32(rng.uniform(-1, 1, n)), f32(rng.uniform(-1, 1, n))), (anyf(), anyf()),
                (f32(rng.uniform(-1, 1, n) * 2.0 ** rng.integers(-60, 60, n)), f32(rng.uniform(-1, 1, n) * 2.0 ** rng.integers(-60, 60, n))),
                (np.nextafter(xx, np.float32(np.inf)) * f32(np.where(rng.random(n) < 0.5, 1, -1)), xx),
                (f32(xx * 2.0 ** rng.integers(-70, -20, n)), xx), (xx, f32(xx * 2.0 ** rng.integers(-70, -20, n))),
                (f32([0.0, -0.0, 0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 1e-45, -1e-45, 3e38] * 12),
                 f32(np.repeat([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 1e-45, -1e-45, 3e38, 1e-30, -2.5], 12)))]
        for y, x in fams:
            a = oracle.libm_f64(2, y.astype(np.float64), x.astype(np.float64))
            for name, want in (("atan2_f32", a.astype(np.float32)), ("atan2_deg_f32", (r2d * a).astype(np.float32))):
                got = djb.selftest_libm(name, y, x, ctx=gpu_ctx).astype(np.float32)
                same = (want.view(np.uint32) == got.view(np.uint32)) | (np.isnan(want) & np.isnan(got))
                assert same.all(), (name, int((~same).sum()), y[~same][:3], x[~same][:3], want[~same][:3], got[~same][:3])
    rng = np.random.default_rng(3)
    n = 1 << 19
    anyf = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32).view(np.float32)
    with np.errstate(all="ignore"):
        for name, code, x, y in (("logf", 0, rng.random(n, dtype=np.float32), None), ("logf", 0, anyf, None),
                                 ("expf", 1, (-30 * rng.random(n)).astype(np.float32), None), ("expf", 1, anyf, None),
                                 ("powf", 2, rng.random(n, dtype=np.float32), (rng.random(n, dtype=np.float32) * 0.6 + 0.45).astype(np.float32)),
                                 ("powf", 2, anyf, rng.permutation(anyf))):
            want = oracle.libm_f32(code, x, y)
            got = djb.selftest_libm(name, x, y, ctx=gpu_ctx).astype(np.float32)
            assert np.array_equal(np.isnan(want), np.isnan(got)), name
            ok = ~np.isnan(want)
            if name == "powf":       # negative bases (glibc's sign_bias path) are left to the device libm: 1 ulp
                neg = x < 0
                assert np.allclose(got[ok & neg], want[ok & neg], rtol=3e-7, atol=0, equal_nan=True)
                ok &= ~neg
            bad = ok & (want.view(np.uint32) != got.view(np.uint32))
            assert not bad.any(), (name, int(bad.sum()), x[bad][:4], None if y is None else y[bad][:4], want[bad][:4], got[bad][:4])


def test_io_hd_roundtrip(gpu_ctx, oracle, dirs):
    i, o, _, _ = dirs
    h, d = djb.brdf.io_to_hd(i, o, ctx=gpu_ctx)
    wh, wd = oracle.io_to_hd(i, o)
    assert_close("h", h, wh); assert_close("d", d, wd)
    gi, go = djb.brdf.hd_to_io(wh, wd, ctx=gpu_ctx)
    wi, wo = oracle.hd_to_io(wh, wd)
    assert_close("i", gi, wi, 1e-5); assert_close("o", go, wo, 2e-5)


def test_merl_index_bit_exact(gpu_ctx, oracle):
    n = 1 << 21
    i = synth.directions_aos(n, synth.SEED_I); o = synth.directions_aos(n, synth.SEED_O)
    got = djb.merl_index(i, o, ctx=gpu_ctx)
    want = oracle.merl_index(i, o)
    mism = int((got != want).sum())
    assert mism == 0, f"{mism} of {n} MERL bin indices differ from the CPU oracle"


def test_merl_eval(gpu_ctx, oracle, dirs):
    i, o, _, _ = dirs
    tab = synth.merl_table_hashed()
    m = djb.merl.from_table(tab, ctx=gpu_ctx)
    om = oracle.merl_from_table(tab)
    for op in ("eval", "evalp", "pdf"):
        got = getattr(m, op)(i, o)
        want = oracle.eval(om, i, o, None, op)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), f"merl {op} not bit-exact"


@pytest.mark.parametrize("case", ["ggx90", "beckmann180", "merl90", "merl90_noshadow", "ggx7"])
def test_fit_tables(gpu_ctx, oracle, case):
    if case.startswith("merl"):
        tab = synth.merl_table(0.3)
        src, osrc = djb.merl.from_table(tab, ctx=gpu_ctx), oracle.merl_from_table(tab)
        res, shadow = 90, case == "merl90"
    elif case == "beckmann180":
        src, osrc = djb.beckmann(None, False, ctx=gpu_ctx), oracle.microfacet("beckmann", ("ideal",), False)
        res, shadow = 180, True
    else:
        src, osrc = djb.ggx(ctx=gpu_ctx), oracle.microfacet("ggx")
        res, shadow = (90 if case == "ggx90" else 7), True
    t = djb.tabular(src, res, shadow, ctx=gpu_ctx)
    want = oracle.tabular_tables(oracle.tabular(osrc, res, shadow))
    got = {"p22": t.get_p22v(), "sigma": t.get_sigmav(), "cdf": t.get_cdfv(), "qf": t.get_qfv(),
           "fresnel": t.get_fresnel().get_points()}
    for k, v in got.items():
        assert_close(f"{case}/{k}", v, want[k], rtol=2e-5)
    ab = djb.tabular.fit_beckmann_parameters(t).get_ellipse()[0]
    ag = djb.tabular.fit_ggx_parameters(t).get_ellipse()[0]
    assert abs(ab - want["alpha_beckmann"]) <= 2e-5 * abs(want["alpha_beckmann"])
    assert abs(ag - want["alpha_ggx"]) <= 2e-5 * abs(want["alpha_ggx"])
    assert "%.3f %.3f" % (ab, ag) == "%.3f %.3f" % (want["alpha_beckmann"], want["alpha_ggx"])
    # the fitted object evaluates like the oracle's
    n = 1 << 14
    i = synth.directions_aos(n, synth.SEED_I); o = synth.directions_aos(n, synth.SEED_O)
    ot = oracle.tabular(osrc, res, shadow)
    for op in ("eval", "pdf"):
        assert_close(f"{case}/tab {op}", getattr(t, op)(i, o), oracle.eval(ot, i, o, None, op), rtol=1e-4)


def test_device_generators_match_numpy(gpu_ctx):
    n = 1 << 16
    d = djb.gen_directions(n, synth.SEED_I, start=12345, ctx=gpu_ctx).cpu().numpy()
    x, y, z = synth.directions(n, synth.SEED_I, start=12345)
    assert np.array_equal(d[0].view(np.uint32), x.view(np.uint32))
    assert np.array_equal(d[1].view(np.uint32), y.view(np.uint32))
    assert np.array_equal(d[2].view(np.uint32), z.view(np.uint32))
    u = djb.gen_uniforms(n, synth.SEED_U1, start=(1 << 33) + 7, ctx=gpu_ctx).cpu().numpy()
    assert np.array_equal(u, synth.rng_uniforms(n, synth.SEED_U1, start=(1 << 33) + 7))


def test_device_tensors_soa_equal_host_aos(gpu_ctx, dirs):
    import torch
    i, o, _, _ = dirs
    g = djb.ggx(djb.fresnel.schlick((1.0, 0.71, 0.29)), True, ctx=gpu_ctx)
    p = djb.microfacet.params.isotropic(0.3)
    host = g.eval(i, o, p)
    ti = torch.from_numpy(i.T.copy()).cuda(); to = torch.from_numpy(o.T.copy()).cuda()
    dev = g.eval(ti, to, p)
    torch.cuda.synchronize()
    assert np.array_equal(dev.cpu().numpy().T.view(np.uint32), host.view(np.uint32))


def test_utia_eval(gpu_ctx, oracle, dirs, tmp_path):
    """utia::eval (16-tap 4-D interpolation + sRGB decode) on a synthetic UTIA-format file."""
    i, o, _, _ = dirs
    rng = np.random.default_rng(11)
    tab = rng.uniform(-5.0, 120.0, size=3 * 288 * 288)     # includes negatives: clamped at load
    p = str(tmp_path / "m.bin"); tab.tofile(p)
    u, ou = djb.utia(p, ctx=gpu_ctx), oracle.utia(p)
    for op in ("eval", "evalp", "pdf"):
        ex = assert_close(f"utia {op}", getattr(u, op)(i, o), oracle.eval(ou, i, o, None, op))
        assert ex > 0.99, f"utia {op}: only {ex:.4f} bit-exact"
    u2 = djb.utia.from_table(tab, ctx=gpu_ctx)
    assert np.array_equal(u2.eval(i, o).view(np.uint32), u.eval(i, o).view(np.uint32))
    with pytest.raises(djb.exc):
        djb.utia(str(tmp_path / "missing.bin"), ctx=gpu_ctx)


def test_lambert_and_default_sampling(gpu_ctx, oracle, dirs):
    i, o, u1, u2 = dirs
    l, ol = djb.lambert(ctx=gpu_ctx), oracle.lambert()
    for op in ("eval", "evalp", "pdf"):
        assert_close(f"lambert {op}", getattr(l, op)(i, o), oracle.eval(ol, i, o, None, op))
    assert_close("lambert sample", l.sample(u1, u2, o), oracle.sample(ol, u1, u2, o), 2e-5)
    w, si, pdf = l.evalp_is(u1, u2, o)
    ww, wi, wpdf = oracle.evalp_is(ol, u1, u2, o)
    assert_close("is pdf", pdf, wpdf, 2e-5); assert_close("is w", w, ww, 4e-5)


def test_standalone_fresnel_eval(gpu_ctx, oracle):
    # fresnel::impl::eval (dj_brdf.h:160) on the five Fresnel classes
    c = np.linspace(0.0, 1.0, 4097).astype(np.float32)
    pts = np.random.default_rng(4).uniform(0, 1, (17, 3)).astype(np.float32)
    for f, fo in ((djb.fresnel.ideal(), ("ideal",)), (djb.fresnel.schlick((1.0, 0.71, 0.29)), ("schlick", 1.0, 0.71, 0.29)),
                  (djb.fresnel.unpolarized((1.5, 2.0, 0.3)), ("unpolarized", 1.5, 2.0, 0.3)),
                  (djb.fresnel.sgd((0.9, 0.5, 0.1), (0.1, 0.2, 0.3)), ("sgd", 0.9, 0.5, 0.1, 0.1, 0.2, 0.3)),
                  (djb.fresnel.spline(pts), ("spline", pts))):
        got = f.eval(c, ctx=gpu_ctx)
        want = oracle.fresnel_eval(oracle.microfacet("ggx", fo, True), c)
        assert_close(f"fresnel::{fo[0]}::eval", got, want, 1e-5)


def test_lambert_reflectance_params(gpu_ctx, oracle, dirs):
    # lambert::params(reflectance) passed as user_param (dj_brdf.h:114-119, 861-868)
    i, o, u1, u2 = dirs
    l, ol = djb.lambert(ctx=gpu_ctx), oracle.lambert()
    lp, op = djb.lambert.params((0.5, 0.25, 0.9)), ("lambert", 0.5, 0.25, 0.9)
    for name in ("eval", "evalp", "pdf"):
        got, want = getattr(l, name)(i, o, lp), oracle.eval(ol, i, o, op, name)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), name
    w, si, pdf = l.evalp_is(u1, u2, o, lp)
    ww, wi, wpdf = oracle.evalp_is(ol, u1, u2, o, op)
    assert_close("lambert(reflectance) evalp_is", w, ww, 2e-5)
    with pytest.raises(djb.exc):
        djb.ggx(ctx=gpu_ctx).eval(i, o, lp)               # lambert::params on a microfacet brdf
    with pytest.raises(djb.exc):
        l.eval(i, o, djb.microfacet.params.isotropic(0.3))  # and the other way round


@pytest.mark.parametrize("ndf", ["ggx", "beckmann"])
def test_microfacet_and_radial_queries(gpu_ctx, oracle, dirs, ndf):
    """microfacet::{ndf,gaf,g1,sigma,p22,vp22,vndf,fresnel} and radial::{p22_radial,...} batched."""
    i, o, u1, u2 = dirs
    fres = ("schlick", 1.0, 0.71, 0.29)
    g = getattr(djb, ndf)(mk_fresnel(fres), True, ctx=gpu_ctx)
    og = oracle.microfacet(ndf, fres, True)
    h = oracle.io_to_hd(i, o)[0]
    for p in (None, ("elliptic", 0.2, 0.5, 0.7), ("pdfparams", 0.4, 0.25, 0.3, 0.1, -0.05)):
        up = mk_params(p)
        assert_close("ndf", g.ndf(h, up), oracle.microfacet_query(og, "ndf", h, params=p))
        assert_close("gaf", g.gaf(h, i, o, up), oracle.microfacet_query(og, "gaf", h, i, o, params=p))
        assert_close("g1", g.g1(h, o, up), oracle.microfacet_query(og, "g1", h, o, params=p))
        assert_close("sigma", g.sigma(o, up), oracle.microfacet_query(og, "sigma", o, params=p))
        assert_close("vndf", g.vndf(h, o, up), oracle.microfacet_query(og, "vndf", h, o, params=p))
        xy = np.stack([u1 * 2 - 1, u2 * 2 - 1, np.zeros_like(u1)], 1).astype(np.float32)
        assert_close("p22", g.p22(xy[:, 0], xy[:, 1], up), oracle.microfacet_query(og, "p22", xy, params=p))
        assert_close("vp22", g.vp22(xy[:, 0], xy[:, 1], o, up), oracle.microfacet_query(og, "vp22", xy, o, params=p))
    c = np.clip(o[:, 2], 0, 1)
    assert_close("fresnel", g.fresnel(c), oracle.fresnel_eval(og, c))
    u = np.clip(u1, 1e-4, 1 - 1e-4)
    s = np.sqrt(1 - c.astype(np.float64) ** 2).astype(np.float32)
    assert_close("p22_radial", g.p22_radial(u * 9), oracle.radial_query(og, "p22_radial", u * 9))
    assert_close("sigma_std_radial", g.sigma_std_radial(c), oracle.radial_query(og, "sigma_std_radial", c))
    assert_close("cdf_radial", g.cdf_radial(u * 5), oracle.radial_query(og, "cdf_radial", u * 5), 2e-5)
    assert_close("qf_radial", g.qf_radial(u), oracle.radial_query(og, "qf_radial", u), 2e-5)
    q3 = g.qf3_radial(u, u * 3 - 1)
    assert_close("qf3_radial", q3, oracle.radial_query(og, "qf3_radial", u, u * 3 - 1), 2e-5)
    q2 = g.qf2_radial(u, c, s)
    want = oracle.radial_query(og, "qf2_radial", u, c, s)
    ok = np.isfinite(want) & (s > 1e-3) & (c > 1e-3)
    assert_close("qf2_radial", q2[ok], want[ok])


def test_tabular_queries_and_not_implemented(gpu_ctx, oracle):
    t = djb.tabular(djb.ggx(ctx=gpu_ctx), 64, True, ctx=gpu_ctx)
    ot = oracle.tabular(oracle.microfacet("ggx"), 64, True)
    u = np.linspace(0.01, 0.99, 500).astype(np.float32)
    assert_close("tab qf_radial", t.qf_radial(u), oracle.radial_query(ot, "qf_radial", u), 1e-4)
    assert_close("tab cdf_radial", t.cdf_radial(u * 4), oracle.radial_query(ot, "cdf_radial", u * 4), 1e-4)
    with pytest.raises(djb.exc) as e:       # radial::qf2_radial base version throws (dj_brdf.h:1854)
        t.qf2_radial(u, u, u)
    assert e.value.status_name == "DJB_ERR_NOT_IMPLEMENTED" and "Not Implemented" in str(e.value)


def test_guarded_fp64_shortcuts_are_exact(gpu_ctx):
    # the rcp/rsq + Newton fast paths must reproduce float(1/sqrt(double x)) and float(1/q) bit for
    # bit; inputs that land near an fp32 rounding boundary take the exact sequence (fallbacks > 0)
    r = djb.selftest_guarded_math(2_000_000_000, seed=20260928, ctx=gpu_ctx)
    assert r["rsqrt_mismatch"] == 0 and r["recip_mismatch"] == 0 and r["srgb_mismatch"] == 0
    # a / b through a double reciprocal (djb_device.hpp fdiv_r; the divisions of mf_p22 by launch-uniform denominators)
    assert r["fdiv_mismatch"] == 0 and r["fdiv_fallback"] > 0
    assert 0 < r["srgb_fallback"] < 2_000_000_000 * 1e-4
    assert 0 < r["rsqrt_fallback"] < 2_000_000_000 * 1e-4
    assert 0 < r["recip_fallback"] < 2_000_000_000 * 1e-4
    # float(sqrt(a)) of a double (1 - c^2: the sin of the stretched view direction, Beckmann's sigma): round 3
    assert r["sqrt_mismatch"] == 0 and 0 < r["sqrt_fallback"] < 2_000_000_000 * 1e-4
    # float(num / den) of two doubles (the one quotient of GGX's quantile function, djb_device.hpp div_to_f32): round 3
    assert r["div_mismatch"] == 0 and 0 < r["div_fallback"] < 2_000_000_000 * 1e-3


def test_microfacet_mutators(gpu_ctx, oracle, dirs):
    # microfacet::set_fresnel / set_shadow (dj_brdf.h:278-279): a mutated handle must behave exactly
    # like one constructed with that state, incl. tab->set_fresnel(fresnel::ideal()) (mitsuba/dj_brdf.cpp:214)
    i, o, _, _ = dirs
    g = djb.ggx(ctx=gpu_ctx)
    g.set_fresnel(djb.fresnel.schlick((1.0, 0.71, 0.29)))
    g.set_shadow(False)
    assert g.get_shadow() == 0
    og = oracle.microfacet("ggx", ("schlick", 1.0, 0.71, 0.29), False)
    got, want = g.eval(i, o), oracle.eval(og, i, o, None, "eval")
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    g.set_fresnel(djb.fresnel.spline(np.array([[0.9, 0.5, 0.1], [0.5, 0.5, 0.5], [1.0, 1.0, 1.0]], np.float32)))
    g.set_shadow(True)
    og = oracle.microfacet("ggx", ("spline", np.array([[0.9, 0.5, 0.1], [0.5, 0.5, 0.5], [1.0, 1.0, 1.0]], np.float32)), True)
    got, want = g.evalp(i, o), oracle.eval(og, i, o, None, "evalp")
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    t = djb.tabular(djb.ggx(ctx=gpu_ctx), 64, True, ctx=gpu_ctx)
    fitted = t.eval(i[:4096], o[:4096])
    t.set_fresnel(djb.fresnel.ideal())
    assert isinstance(t.get_fresnel(), djb.fresnel.ideal)
    ideal = t.eval(i[:4096], o[:4096])
    assert np.isfinite(ideal).all() and not np.array_equal(ideal, fitted)
    from dj_brdf_amd import _lib
    lam = djb.lambert(ctx=gpu_ctx)
    with pytest.raises(djb.exc):                     # not a microfacet BRDF
        _lib.check(_lib.load().djb_brdf_set_shadow(lam._h, 1))


def test_host_views_of_any_layout(gpu_ctx, dirs):
    # DJB_MEM_HOST copies the caller's own layout (interleaved vec3, contiguous SoA, three separate
    # component arrays) and only packs exotic strides (here: padded float4 records); all must agree
    import ctypes as C
    from dj_brdf_amd import _lib
    lib = _lib.load()
    i, o, _, _ = dirs
    n = 10_007
    i, o = np.ascontiguousarray(i[:n]), np.ascontiguousarray(o[:n])
    g = djb.ggx(djb.fresnel.schlick((1.0, 0.71, 0.29)), True, ctx=gpu_ctx)
    want = g.eval(i, o)                                            # interleaved in, interleaved out

    def view(x, y, z, stride):
        v = _lib.Vec3View(); v.x, v.y, v.z, v.stride = x, y, z, stride
        return v

    def run(vi, vo, vout):
        _lib.check(lib.djb_eval_batch(gpu_ctx._h, g._h, C.c_int64(n), C.byref(vi), C.byref(vo), None, C.byref(vout),
                                      C.c_int(_lib.MEM_HOST)))

    # three separately allocated component arrays (SoA, not contiguous with each other)
    ic = [np.ascontiguousarray(i[:, c]) for c in range(3)]; oc = [np.ascontiguousarray(o[:, c]) for c in range(3)]
    rc = [np.full(n, np.nan, np.float32) for _ in range(3)]
    run(view(*[a.ctypes.data for a in ic], 1), view(*[a.ctypes.data for a in oc], 1), view(*[a.ctypes.data for a in rc], 1))
    assert np.array_equal(np.stack(rc, 1).view(np.uint32), want.view(np.uint32))
    # padded float4 records (stride 4): the packed fallback, input and output
    i4 = np.zeros((n, 4), np.float32); i4[:, :3] = i
    o4 = np.zeros((n, 4), np.float32); o4[:, :3] = o
    r4 = np.full((n, 4), 7.0, np.float32)
    b = lambda a: a.ctypes.data
    run(view(b(i4), b(i4) + 4, b(i4) + 8, 4), view(b(o4), b(o4) + 4, b(o4) + 8, 4), view(b(r4), b(r4) + 4, b(r4) + 8, 4))
    assert np.array_equal(r4[:, :3].view(np.uint32), want.view(np.uint32)) and (r4[:, 3] == 7.0).all()
    # mixed: SoA in, interleaved out
    r = np.empty((n, 3), np.float32)
    run(view(*[a.ctypes.data for a in ic], 1), view(b(o4), b(o4) + 4, b(o4) + 8, 4), view(b(r), b(r) + 4, b(r) + 8, 3))
    assert np.array_equal(r.view(np.uint32), want.view(np.uint32))


def test_large_host_batches_are_chunked_with_identical_results(gpu_ctx, dirs, monkeypatch):
    # DJB_MEM_HOST batches of >= 2 chunks are copied in / out chunk by chunk with both PCIe directions
    # in flight (eval_host_pipelined); the chunk size is forced down so a few thousand pairs exercise
    # it: 5 chunks with a ragged tail, every layout the path accepts, eval / evalp / pdf / eval+pdf,
    # an analytic lobe and the two-tier MERL lookup.  Expected = the same call with chunking off.
    import ctypes as C
    from dj_brdf_amd import _lib
    lib = _lib.load()
    i, o, _, _ = dirs
    n = 43_211
    i, o = np.ascontiguousarray(i[:n]), np.ascontiguousarray(o[:n])
    g = djb.ggx(djb.fresnel.schlick((1.0, 0.71, 0.29)), True, ctx=gpu_ctx)
    m = djb.merl.from_table(synth.merl_table_hashed(), ctx=gpu_ctx)

    def view(a, aos):
        v = _lib.Vec3View(); b = a.ctypes.data
        if aos: v.x, v.y, v.z, v.stride = b, b + 4, b + 8, 3
        else: v.x, v.y, v.z, v.stride = b, b + 4 * n, b + 8 * n, 1
        return v

    def paged(shape, fill=None):
        # a buffer that owns its host pages (the chunked path is only taken when inputs and outputs share
        # none; small numpy arrays come from the malloc heap and may)
        count = int(np.prod(shape))
        raw = np.empty(count + 2 * 1024 + 1024, np.float32)
        off = (-raw.ctypes.data % 4096) // 4
        a = raw[off:off + count].reshape(shape)
        assert a.ctypes.data % 4096 == 0
        a[...] = np.nan if fill is None else fill
        return a

    i, o = paged(i.shape, i), paged(o.shape, o)
    it, ot = paged((3, n), i.T), paged((3, n), o.T)

    def run(brdf, aos_in, aos_out, chunk):
        monkeypatch.setenv("DJB_HOST_PIPE_CHUNK", str(chunk))
        if chunk: monkeypatch.setenv("DJB_HOST_PIPE_REQUIRE", "1")    # falling back to the plain path is an error here
        else: monkeypatch.delenv("DJB_HOST_PIPE_REQUIRE", raising=False)
        ii = i if aos_in else it; oo = o if aos_in else ot
        res = {}
        for name, fn in (("eval", lib.djb_eval_batch), ("evalp", lib.djb_evalp_batch)):
            out = paged((n, 3) if aos_out else (3, n))
            _lib.check(fn(gpu_ctx._h, brdf._h, C.c_int64(n), C.byref(view(ii, aos_in)), C.byref(view(oo, aos_in)), None,
                          C.byref(view(out, aos_out)), C.c_int(_lib.MEM_HOST)))
            res[name] = out if aos_out else out.T
        pdf = paged((n,))
        _lib.check(lib.djb_pdf_batch(gpu_ctx._h, brdf._h, C.c_int64(n), C.byref(view(ii, aos_in)), C.byref(view(oo, aos_in)), None,
                                     pdf.ctypes.data_as(C.POINTER(C.c_float)), C.c_int(_lib.MEM_HOST)))
        res["pdf"] = pdf
        out = paged((n, 3) if aos_out else (3, n)); pdf2 = paged((n,))
        _lib.check(lib.djb_eval_pdf_batch(gpu_ctx._h, brdf._h, C.c_int64(n), C.byref(view(ii, aos_in)), C.byref(view(oo, aos_in)), None,
                                          C.c_int(1), C.byref(view(out, aos_out)), pdf2.ctypes.data_as(C.POINTER(C.c_float)),
                                          C.c_int(_lib.MEM_HOST)))
        res["evalp+pdf"] = np.concatenate([out if aos_out else out.T, pdf2[:, None]], 1)
        return res

    for brdf in (g, m):
        want = run(brdf, True, True, 0)
        assert all(np.isfinite(v).all() for v in want.values())
        for aos_in, aos_out, chunk in ((True, True, 10_000), (False, False, 10_000), (True, False, 9_001), (False, True, 21_605)):
            got = run(brdf, aos_in, aos_out, chunk)
            for k in want:
                assert np.array_equal(got[k].view(np.uint32), want[k].view(np.uint32)), (k, aos_in, aos_out, chunk)
    # sample / evalp_is: 20 B in (u1, u2, o), 12 or 28 B out (i [, weight, pdf]) per unit
    u1, u2 = paged((n,), dirs[2][:n]), paged((n,), dirs[3][:n])
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))

    def run_s(brdf, aos, chunk):
        monkeypatch.setenv("DJB_HOST_PIPE_CHUNK", str(chunk))
        if chunk: monkeypatch.setenv("DJB_HOST_PIPE_REQUIRE", "1")
        else: monkeypatch.delenv("DJB_HOST_PIPE_REQUIRE", raising=False)
        oo = o if aos else ot
        shp = (n, 3) if aos else (3, n)
        si = paged(shp)
        _lib.check(lib.djb_sample_batch(gpu_ctx._h, brdf._h, C.c_int64(n), fp(u1), fp(u2), C.byref(view(oo, aos)), None,
                                        C.byref(view(si, aos)), C.c_int(_lib.MEM_HOST)))
        wi, ww, wp = paged(shp), paged(shp), paged((n,))
        _lib.check(lib.djb_evalp_is_batch(gpu_ctx._h, brdf._h, C.c_int64(n), fp(u1), fp(u2), C.byref(view(oo, aos)), None,
                                          C.byref(view(ww, aos)), C.byref(view(wi, aos)), fp(wp), C.c_int(_lib.MEM_HOST)))
        t = (lambda a: a) if aos else (lambda a: a.T)
        return {"sample": t(si), "is.i": t(wi), "is.w": t(ww), "is.pdf": wp}

    for brdf in (g, djb.beckmann(ctx=gpu_ctx)):
        want = run_s(brdf, True, 0)
        assert all(np.isfinite(v).all() for v in want.values())
        for aos, chunk in ((True, 10_000), (False, 12_345)):
            got = run_s(brdf, aos, chunk)
            for k in want:
                assert np.array_equal(got[k].view(np.uint32), want[k].view(np.uint32)), (k, aos, chunk)
    # device-resident answers agree as well (the chunk kernels are the plain kernels)
    import torch
    ti, to = torch.from_numpy(i).cuda(), torch.from_numpy(o).cuda()
    monkeypatch.setenv("DJB_HOST_PIPE_CHUNK", "10000")
    monkeypatch.delenv("DJB_HOST_PIPE_REQUIRE", raising=False)
    assert np.array_equal(m.eval(i, o).view(np.uint32), m.eval(ti, to).cpu().numpy().view(np.uint32))


def test_concurrent_callers_share_one_context(gpu_ctx):
    # the reference's operators are const and thread-safe (Mitsuba render threads share one BSDF):
    # concurrent host threads on ONE context must get the same bits as sequential calls, including the
    # two-tier MERL lookup whose three launches share per-context scratch
    import threading
    m = djb.merl.from_table(synth.merl_table_hashed(), ctx=gpu_ctx)
    g = djb.ggx(djb.fresnel.schlick((1.0, 0.71, 0.29)), True, ctx=gpu_ctx)
    n, T = 300_000, 6
    ins = [(synth.directions_aos(n, synth.SEED_I + 7 * t), synth.directions_aos(n, synth.SEED_O + 13 * t)) for t in range(T)]
    want = [(m.eval(i, o), g.eval(i, o)) for i, o in ins]
    got, errs = [None] * T, []

    def work(t):
        try:
            for _ in range(3):
                got[t] = (m.eval(*ins[t]), g.eval(*ins[t]))
        except Exception as e:   # surfaced below
            errs.append(e)

    ths = [threading.Thread(target=work, args=(t,)) for t in range(T)]
    for t in ths: t.start()
    for t in ths: t.join()
    assert not errs, errs
    for t in range(T):
        assert np.array_equal(got[t][0].view(np.uint32), want[t][0].view(np.uint32)), f"thread {t}: merl"
        assert np.array_equal(got[t][1].view(np.uint32), want[t][1].view(np.uint32)), f"thread {t}: ggx"


def test_get_samples(gpu_ctx, tmp_path):
    # merl::get_samples / utia::get_samples (dj_brdf.h:132, 143)
    tab = synth.merl_table_hashed()
    m = djb.merl.from_table(tab, ctx=gpu_ctx)
    assert np.array_equal(m.get_samples(), np.ascontiguousarray(tab, np.float64).reshape(-1))
    p = str(tmp_path / "m.binary"); synth.write_merl_binary(p, tab)
    assert np.array_equal(djb.merl(p, ctx=gpu_ctx).get_samples(), m.get_samples())
    raw = np.random.default_rng(3).uniform(-5.0, 120.0, size=3 * 288 * 288)
    u = djb.utia.from_table(raw, ctx=gpu_ctx)
    want = np.maximum(0.0, raw) * np.float64(np.float32(1.0) / np.float32(140.0))     # utia::normalize, dj_brdf.h:1162-1177
    assert np.array_equal(u.get_samples(), want)
    with pytest.raises(djb.exc):
        from dj_brdf_amd import _lib
        import ctypes as C
        n = C.c_int64(0)
        g = djb.ggx(ctx=gpu_ctx)          # kept alive across the call: `djb.ggx(...)._h` alone hands a freed handle to the library
        _lib.check(_lib.load().djb_brdf_get_samples(g._h, None, C.c_int64(0), C.byref(n)))


SHARP_PARAMS = [("elliptic", 0.05, 0.05, 0.0), ("elliptic", 0.02, 0.1, 0.3), ("pdfparams", 0.05, 0.08, 0.5, 0.0, 0.0), ("elliptic", 0.001, 0.1, 1.0)]


def hostile_pairs(dirs):
    """the bench directions with families a renderer can produce mixed in: either direction below the horizon, un-normalised, pointing away
    from each other, grazing, on the normal, equal, NaN / Inf / zero vectors, h on the normal"""
    i, o, _, _ = dirs
    i, o = i.copy(), o.copy()
    n = i.shape[0]; k = n // 16
    o[:k, 2] *= -1                                                   # o below the horizon
    i[k:2 * k, 2] *= -1                                              # i below the horizon
    i[2 * k:3 * k] *= 0.01; o[2 * k:3 * k] *= 6.0                   # un-normalised, very different lengths (dot(i, h) may turn negative)
    i[3 * k:3 * k + k // 2, 0] *= -5.0                               # i pointing away from o: dot(i, h) < 0
    o[4 * k:5 * k, 2] = 2e-4 * np.abs(o[4 * k:5 * k, 2])             # grazing o (around the z > 1e-4 cut)
    i[5 * k:6 * k] = np.array([0, 0, 1], np.float32) + 1e-4 * i[5 * k:6 * k]     # on the normal: sigma's pole
    o[6 * k:7 * k] = i[6 * k:7 * k]                                  # i == o
    o[7 * k:7 * k + 64, 0] = np.nan; i[7 * k + 64:7 * k + 128, 2] = np.inf; o[7 * k + 128:7 * k + 192] = 0.0; i[7 * k + 192:7 * k + 256, 1] = -np.inf
    i[7 * k + 256:7 * k + 320, 2] = 0.0; o[7 * k + 320:7 * k + 384, 2] = 0.0; i[7 * k + 384:7 * k + 448, 2] = -0.0      # exactly on the horizon
    # a NaN in x or y of ONE direction with both z's fine: no shadowing -> G = g1(o) > 0, F(sat(NaN)) = NaN for Schlick / unpolarized,
    # and the reference returns NaN * 0 = NaN (a max() over the components would hide the NaN: ADVICE r04)
    i[7 * k + 448:7 * k + 512, 0] = np.nan; i[7 * k + 512:7 * k + 576, 1] = np.nan; o[7 * k + 576:7 * k + 640, 1] = np.nan
    o[8 * k:9 * k, :2] *= 1e-3                                       # h nearly on the normal: r^2 small even for a sharp lobe
    i[9 * k:10 * k] = -o[9 * k:10 * k] * np.array([1, 1, -1], np.float32) * 3.0      # mirror pairs of different lengths: h on the normal
    return i, o


def value_bits(a):
    """the bits of every value, signs of zeros included; NaNs (whose payload is the processor's business) as one pattern"""
    a = np.ascontiguousarray(a, np.float32)
    return np.where(np.isnan(a), np.uint32(0x7fc00000), a.view(np.uint32))


@pytest.mark.parametrize("ndf", ["ggx", "beckmann"])
def test_microfacet_eval_on_hostile_pairs(gpu_ctx, oracle, dirs, ndf):
    """eval = evalp / i.z divides evalp's vec3(0) all the same (dj_brdf.h:1551-1555): -0 for i below the horizon, NaN on it -- and everything
    else a renderer's stray pairs produce: the kernels return the reference's bits (NaN payloads aside), every output set, on the device
    and on the product's host path"""
    import torch
    i, o = hostile_pairs(dirs)
    di, do = torch.from_numpy(np.ascontiguousarray(i.T)).cuda(), torch.from_numpy(np.ascontiguousarray(o.T)).cuda()
    cpu = djb.cpu_context()
    m = 1 << 13                                                       # the host path on a slice (every family is 2^13 long)
    sl = np.r_[0:64, 8192:8192 + 64, 7 * 8192:7 * 8192 + 448, 3 * 8192:3 * 8192 + 64]
    for fres in (("ideal",), ("schlick", 1.0, 0.71, 0.29)):
        for shadow in (True, False):
            g = getattr(djb, ndf)(mk_fresnel(fres), shadow, ctx=gpu_ctx); gc = getattr(djb, ndf)(mk_fresnel(fres), shadow, ctx=cpu)
            ob = oracle.microfacet(ndf, fres, shadow)
            for p in (None, ("elliptic", 0.3, 0.3, 0.0), ("pdfparams", 0.4, 0.25, 0.3, 0.1, -0.05)):
                up = mk_params(p)
                for op in ("eval", "evalp", "pdf"):
                    want = oracle.eval(ob, i, o, p, op)
                    got = getattr(g, op)(di, do, up).cpu().numpy()
                    got = got.T if got.ndim == 2 else got
                    assert np.array_equal(value_bits(got), value_bits(want)), (ndf, fres, shadow, p, op, int(np.sum(value_bits(got) != value_bits(want))))
                    hc = np.asarray(getattr(gc, op)(i[sl], o[sl], up))
                    assert np.array_equal(value_bits(hc), value_bits(want[sl])), (ndf, fres, shadow, p, op, "host path")
    del m


@pytest.mark.parametrize("fres", [("ideal",), ("schlick", 1.0, 0.71, 0.29), ("unpolarized", 1.5, 1.8, 2.4), ("schlick", 0.0, 1.0, 0.5)], ids=lambda f: "-".join(str(x) for x in f))
def test_beckmann_sharp_lobe_two_path_kernel(gpu_ctx, oracle, dirs, fres):
    """k_eval_bk_sharp (lobes with alpha <= 0.1: pairs whose result is a known zero are written at once, the others evaluated in dense
    waves) must return the reference's BITS -- zeros with their signs, NaNs, everything: against the oracle on the bench directions
    mixed with below-horizon, un-normalised, grazing, on-the-normal, i == o, NaN and Inf pairs; every output set, both layouts."""
    import torch
    i, o = hostile_pairs(dirs)
    bits = value_bits
    di, do = torch.from_numpy(np.ascontiguousarray(i.T)).cuda(), torch.from_numpy(np.ascontiguousarray(o.T)).cuda()      # dense SoA
    ai, ao = torch.from_numpy(i).cuda(), torch.from_numpy(o).cuda()                                                      # array of vec3
    trivial_share = []
    for shadow in (True, False):
        g = djb.beckmann(mk_fresnel(fres), shadow, ctx=gpu_ctx)
        ob = oracle.microfacet("beckmann", fres, shadow)
        for p in SHARP_PARAMS:
            up = mk_params(p)
            want = {op: oracle.eval(ob, i, o, p, op) for op in ("eval", "evalp", "pdf")}
            trivial_share.append(float(np.mean(np.all(want["eval"] == 0, axis=1))))
            for op in ("eval", "evalp", "pdf"):
                got_s = getattr(g, op)(di, do, up).cpu().numpy(); got_a = getattr(g, op)(ai, ao, up).cpu().numpy()
                got_s = got_s.T if got_s.ndim == 2 else got_s
                assert np.array_equal(bits(got_s), bits(want[op])), (fres, shadow, p, op, "soa", int(np.sum(bits(got_s) != bits(want[op]))))
                assert np.array_equal(bits(got_a), bits(want[op])), (fres, shadow, p, op, "aos")
            for cos in (False, True):
                fr, pdf = g.eval_pdf(di, do, up, cos=cos)
                assert np.array_equal(bits(fr.cpu().numpy().T), bits(want["evalp" if cos else "eval"])) and np.array_equal(bits(pdf.cpu().numpy()), bits(want["pdf"])), (fres, shadow, p, cos)
    assert max(trivial_share) > 0.5, "the test lobes are not sharp enough to exercise the trivial path"


def test_every_kind_and_operator_on_hostile_pairs(gpu_ctx):
    """tools/hostile_parity_sweep.py: ten BRDF kinds x (eval, evalp, pdf, sample, evalp_is) on the hostile pairs and uniforms outside [0, 1) /
    NaN, device kernels against the oracle, value bits with the signs of zeros"""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "hostile_parity_sweep.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert lines[-1] == "cases with a mismatch: 0" and sum(l.endswith(" ok") for l in lines) >= 240, "\n".join(l for l in lines if not l.endswith(" ok"))
