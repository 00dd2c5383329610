"""DJB_OPT_CONTRACT_1E5 (csrc/djb_kernels_contract.hip): GGX eval / evalp / pdf inside the north star's VALUE contract
-- every result within 1e-5 relative of the reference's, zeros exactly where the reference returns zeros -- instead of
bit-identically.  The tolerance asserted here is the contract itself: RTOL = 1e-5 (north_star), measured maxima are
printed.  The option is off by default; nothing else in the suite runs with it.
"""
import numpy as np
import pytest

from dj_brdf_amd import djb, synth

pytestmark = pytest.mark.gpu

RTOL = 1e-5
N = 1 << 18
PARAMS = [None, ("elliptic", 0.3, 0.3, 0.0), ("elliptic", 0.2, 0.5, 0.7), ("elliptic", 0.05, 0.05, 0.0),
          ("pdfparams", 0.4, 0.25, 0.3, 0.0, 0.0)]


def mk_params(p):
    if p is None: return None
    if p[0] == "elliptic": return djb.microfacet.params.elliptic(*p[1:])
    return djb.microfacet.params.pdfparams(*p[1:])


def soa(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a.T)).cuda()      # [3, n] device tensor = dense SoA views


def check_contract(name, got, want):
    """zeros / NaNs exactly where the reference has them; everything else within RTOL relative"""
    got = np.asarray(got, np.float64); want = np.asarray(want, np.float64)
    assert got.shape == want.shape, name
    assert np.array_equal(np.isnan(got), np.isnan(want)), f"{name}: NaN pattern differs"
    assert np.array_equal(got == 0, want == 0), f"{name}: zero pattern differs ({np.sum((got == 0) != (want == 0))} values)"
    m = np.isfinite(want) & (want != 0)
    rel = np.abs(got[m] - want[m]) / np.abs(want[m])
    mx = float(rel.max()) if rel.size else 0.0
    assert mx <= RTOL, f"{name}: max relative error {mx:.3e} outside the {RTOL} contract"
    return mx


@pytest.fixture()
def ct_ctx(gpu_ctx):
    djb.set_contract_1e5(gpu_ctx, True)
    yield gpu_ctx
    djb.set_contract_1e5(gpu_ctx, False)


def mk_fresnel(fres):
    if fres[0] == "ideal": return djb.fresnel.ideal()
    if fres[0] == "schlick": return djb.fresnel.schlick(fres[1:])
    return djb.fresnel.unpolarized(fres[1:])


@pytest.mark.parametrize("ndf", ["ggx", "beckmann"])
@pytest.mark.parametrize("fres", [("ideal",), ("schlick", 1.0, 0.71, 0.29), ("unpolarized", 1.5, 1.8, 2.4), ("unpolarized", 1.05, 1.33, 40.0)],
                         ids=lambda f: "-".join(str(x) for x in f))
def test_contract_lobes_vs_oracle(ct_ctx, oracle, fres, ndf):
    i, o = synth.directions_aos(N, synth.SEED_I), synth.directions_aos(N, synth.SEED_O)
    di, do = soa(i), soa(o)
    worst = 0.0
    for shadow in (True, False):
        g = getattr(djb, ndf)(mk_fresnel(fres), shadow, ctx=ct_ctx)
        ob = oracle.microfacet(ndf, fres, shadow)
        for p in PARAMS:
            up = mk_params(p)
            differs = 0
            for op in ("eval", "evalp", "pdf"):
                got = getattr(g, op)(di, do, up).cpu().numpy()
                got = got.T if got.ndim == 2 else got
                want = oracle.eval(ob, i, o, p, op)
                worst = max(worst, check_contract(f"{ndf}/{fres[0]}/{shadow}/{p}/{op}", got, want))
                differs += int(np.sum(np.ascontiguousarray(got, np.float32).view(np.uint32) != want.view(np.uint32)))
            fr, pdf = g.eval_pdf(di, do, up)
            worst = max(worst, check_contract("fused eval", fr.cpu().numpy().T, oracle.eval(ob, i, o, p, "eval")))
            worst = max(worst, check_contract("fused pdf", pdf.cpu().numpy(), oracle.eval(ob, i, o, p, "pdf")))
            # evidence that the fast path, not the bit-exact kernel, produced these values
            # (Beckmann at alpha = 0.05: exp(-r^2) sends most of the batch into the denormal tail, i.e. to the exact tier)
            if not (ndf == "beckmann" and p == ("elliptic", 0.05, 0.05, 0.0)):
                assert differs > 0, "contract mode returned bit-identical values everywhere: the fast path did not run"
    print(f"\ncontract mode {ndf}/{fres[0]}: max relative error vs the oracle {worst:.3e} (contract {RTOL})")


def test_contract_off_and_outside_domain_is_bit_exact(gpu_ctx, oracle):
    """off by default; and with the option on, set-ups outside the fast path's domain (mean-normal offset, other Fresnel
    terms, host / strided layouts) still take the bit-exact kernels"""
    i, o = synth.directions_aos(1 << 14, synth.SEED_I), synth.directions_aos(1 << 14, synth.SEED_O)
    bits = lambda a: np.ascontiguousarray(a, np.float32).view(np.uint32)
    g = djb.ggx(djb.fresnel.ideal(), True, ctx=gpu_ctx)
    ob = oracle.microfacet("ggx", ("ideal",), True)
    p = ("elliptic", 0.3, 0.3, 0.0)
    assert np.array_equal(bits(g.eval(soa(i), soa(o), mk_params(p)).cpu().numpy().T), bits(oracle.eval(ob, i, o, p, "eval")))
    djb.set_contract_1e5(gpu_ctx, True)
    try:
        q = ("pdfparams", 0.4, 0.25, 0.3, 0.1, -0.05)            # offset lobe: outside the domain
        assert np.array_equal(bits(g.eval(soa(i), soa(o), mk_params(q)).cpu().numpy().T), bits(oracle.eval(ob, i, o, q, "eval")))
        assert np.array_equal(bits(g.eval(i, o, mk_params(p))), bits(oracle.eval(ob, i, o, p, "eval")))   # host AoS batch
        # an index of refraction below 1.05: the reference's own g - c is noisier than the contract there (ct_unpolarized)
        gu = djb.ggx(djb.fresnel.unpolarized((1.5, 1.02, 2.4)), True, ctx=gpu_ctx)
        ou = oracle.microfacet("ggx", ("unpolarized", 1.5, 1.02, 2.4), True)
        assert np.array_equal(bits(gu.eval(soa(i), soa(o), mk_params(p)).cpu().numpy().T), bits(oracle.eval(ou, i, o, p, "eval")))
        t = djb.tabular(djb.ggx(ctx=gpu_ctx), 32, True, ctx=gpu_ctx)       # fitted lobes are outside the domain
        ot = oracle.tabular(oracle.microfacet("ggx"), 32, True)
        assert np.array_equal(bits(t.eval(soa(i), soa(o)).cpu().numpy().T), bits(oracle.eval(ot, i, o, None, "eval")))
    finally:
        djb.set_contract_1e5(gpu_ctx, False)


@pytest.mark.parametrize("ndf", ["ggx", "beckmann"])
def test_contract_hostile_inputs_match_the_exact_kernels(ct_ctx, ndf):
    """below-horizon, grazing, opposite, zero, un-normalised, NaN / Inf components, ragged tail (n % 4 != 0): the
    two-tier result has the exact kernels' zeros and NaNs and stays within the contract everywhere else"""
    rng = np.random.default_rng(5)
    n = (1 << 16) + 3
    i = synth.directions_aos(n, 11).copy(); o = synth.directions_aos(n, 12).copy()
    k = n // 8
    o[:k, 2] *= -1                                            # below the horizon
    i[k:2 * k, 2] *= 1e-4; i[k:2 * k] /= np.linalg.norm(i[k:2 * k], axis=1, keepdims=True)       # grazing
    o[2 * k:3 * k] = i[2 * k:3 * k] * np.array([-1, -1, 1], np.float32)                             # mirror pairs
    o[3 * k:4 * k] = i[3 * k:4 * k]                                                                  # identical
    i[4 * k:5 * k] *= rng.uniform(0.1, 10, (k, 1)).astype(np.float32)                               # un-normalised
    o[5 * k:5 * k + 8] = 0
    i[5 * k + 8:5 * k + 16, 0] = np.nan
    o[5 * k + 16:5 * k + 24, 1] = np.inf
    o[5 * k + 24:5 * k + 32, 2] = 1e-20
    i[6 * k:7 * k, :2] *= 1e-3; i[6 * k:7 * k] /= np.linalg.norm(i[6 * k:7 * k], axis=1, keepdims=True)   # near-normal
    i = i.astype(np.float32); o = o.astype(np.float32)
    p = djb.microfacet.params.elliptic(0.2, 0.5, 0.7)
    di, do = soa(i), soa(o)
    res = []
    for fres in (djb.fresnel.schlick((1.0, 0.71, 0.29)), djb.fresnel.unpolarized((1.5, 1.05, 2.4))):
        g = getattr(djb, ndf)(fres, True, ctx=ct_ctx)
        fr, pdf = g.eval_pdf(di, do, p)
        djb.set_contract_1e5(ct_ctx, False)
        fe, pe = g.eval_pdf(di, do, p)
        djb.set_contract_1e5(ct_ctx, True)
        res += [(ndf + " eval", fr, fe), (ndf + " pdf", pdf, pe)]
    with np.errstate(all="ignore"):
        for name, a, b in res:
            a = a.cpu().numpy().astype(np.float64); b = b.cpu().numpy().astype(np.float64)
            assert np.array_equal(np.isnan(a), np.isnan(b)), f"{name}: NaN pattern"
            assert np.array_equal(np.isinf(a), np.isinf(b)) and np.array_equal(a[np.isinf(b)], b[np.isinf(b)]), f"{name}: Inf pattern"
            assert np.array_equal(a == 0, b == 0), f"{name}: zero pattern"
            m = np.isfinite(b) & (b != 0)
            rel = np.abs(a[m] - b[m]) / np.abs(b[m])
            assert rel.max() <= RTOL, f"{name}: {rel.max():.3e}"


@pytest.mark.parametrize("ndf", ["ggx", "beckmann"])
@pytest.mark.parametrize("family", [0, 1, 2, 3, 4])
def test_contract_selftest_families(gpu_ctx, family, ndf):
    """fast path vs the bit-exact per-pair code on 2^26 generated pairs per set-up, on the device"""
    worst_e = worst_p = 0.0
    t2 = 0
    for fres, p in ((djb.fresnel.ideal(), djb.microfacet.params.isotropic(0.3)),
                    (djb.fresnel.schlick((1.0, 0.71, 0.29)), djb.microfacet.params.elliptic(0.2, 0.5, 0.7)),
                    (djb.fresnel.schlick((0.04, 0.04, 0.04)), djb.microfacet.params.isotropic(0.05)),
                    (djb.fresnel.unpolarized((1.05, 1.5, 300.0)), djb.microfacet.params.elliptic(0.3, 0.6, 0.2)),
                    (djb.fresnel.ideal(), djb.microfacet.params.pdfparams(0.9, 0.1, 0.89))):
        g = getattr(djb, ndf)(fres, True, ctx=gpu_ctx)
        r = djb.selftest_contract(g, p, n=1 << 26, seed=77 + family, family=family, ctx=gpu_ctx)
        assert r["pairs"] == 1 << 26
        assert r["zero_mismatch"] == 0 and r["outside_1e5"] == 0, r
        worst_e, worst_p, t2 = max(worst_e, r["max_rel_eval"]), max(worst_p, r["max_rel_pdf"]), max(t2, r["tier2"])
    print(f"\ncontract selftest {ndf} family {family}: max rel eval {worst_e:.3e}, pdf {worst_p:.3e}, tier-2 share <= {t2 / (1 << 26):.2e}")
    assert worst_e <= RTOL and worst_p <= RTOL


def test_contract_random_setups(gpu_ctx):
    """fuzz: 60 random lobes inside the fast path's domain (roughness 0.05 .. 2 per axis, correlation up to the 0.9 limit,
    ideal or Schlick Fresnel with f0 down to the 0.01 limit, shadowing on / off) x the five input families, 2^22 generated
    pairs each: tier 1 against the bit-exact per-pair code on the device"""
    rng = np.random.default_rng(20260928)
    worst = 0.0
    for k in range(60):
        ndf = ("ggx", "beckmann")[k % 2]
        ax, ay = (float(v) for v in 10.0 ** rng.uniform(-1.3, 0.3, 2))
        rho = float(rng.uniform(-0.9, 0.9)) if k % 3 else 0.0
        fres = djb.fresnel.ideal() if k % 4 == 0 else djb.fresnel.schlick(tuple(float(v) for v in rng.uniform(0.01, 1.0, 3))) if k % 4 != 3 \
            else djb.fresnel.unpolarized(tuple(float(v) for v in 1.05 * 10.0 ** rng.uniform(0.0, 2.0, 3)))
        g = getattr(djb, ndf)(fres, bool(k % 5), ctx=gpu_ctx)
        p = djb.microfacet.params.pdfparams(ax, ay, rho)
        r = djb.selftest_contract(g, p, n=1 << 22, seed=1000 + k, family=k % 5, ctx=gpu_ctx)
        assert r["zero_mismatch"] == 0 and r["outside_1e5"] == 0, (ndf, ax, ay, rho, r)
        worst = max(worst, r["max_rel_eval"], r["max_rel_pdf"])
    print(f"\ncontract fuzz: worst relative difference over 60 random set-ups {worst:.3e}")
    assert worst <= RTOL


# ---------------------------------------------------------------------------------------------- ABC (abc::eval)
def test_contract_abc_all_materials_selftest(gpu_ctx):
    """The ABC fast path (k_ct_fast_v4<ABC>) against the bit-exact per-pair code on the device: the 100 published rows x
    the five input families, 2^22 generated pairs each (B reaches 8e6, C 2.7, ior 1.04 .. 100)."""
    worst, t2 = 0.0, 0
    for k, name in enumerate(synth.MERL_NAMES):
        b = djb.abc(name, ctx=gpu_ctx)
        for family in range(5):
            r = djb.selftest_contract(b, None, n=1 << 22, seed=300 + 5 * k + family, family=family, ctx=gpu_ctx)
            assert r["pairs"] == 1 << 22
            assert r["zero_mismatch"] == 0 and r["outside_1e5"] == 0, (name, family, r)
            worst, t2 = max(worst, r["max_rel_eval"], r["max_rel_pdf"]), max(t2, r["tier2"])
    print(f"\ncontract abc: worst relative difference over 100 materials x 5 families {worst:.3e}, tier-2 share <= {t2 / (1 << 22):.2e}")
    assert worst <= RTOL


def test_contract_abc_vs_oracle_and_domain(ct_ctx, oracle):
    """eval / evalp / pdf / fused through the batch API (dense device views -> the two-tier kernels) against the oracle,
    hostile inputs included; rows outside the fast path's domain (ior <= 1, an exponent beyond 16) and sgd objects stay
    on the bit-exact kernels."""
    from dj_brdf_amd import param_tables
    rng = np.random.default_rng(5)
    n = N                                                         # rows of a [3, n] tensor stay 16-byte aligned: the two-tier kernels run
    i, o = synth.directions_aos(n, synth.SEED_I), synth.directions_aos(n, synth.SEED_O)
    i[:2000, 2] *= -1; o[2000:4000, 2] *= -1                      # below the horizon -> zeros
    i[4000:4100] = np.nan; o[4100:4200, 0] = np.inf
    i[4200:6200] = o[4200:6200] * np.float32([-1, -1, 1]) + rng.normal(0, 1e-4, (2000, 3)).astype(np.float32)   # theta_d -> 90 deg
    o[6200:8200] = i[6200:8200]                                   # h = i = o
    i[8200:9200] *= np.float32(3.0)                               # un-normalised
    di, do = soa(i), soa(o)
    worst = 0.0
    for name in ("gold-metallic-paint", "alum-bronze", "black-fabric", "chrome", "white-marble", "yellow-plastic", "pearl-paint", "teflon"):
        b, ob = djb.abc(name, ctx=ct_ctx), oracle.abc(name)
        for mode in ("eval", "evalp", "pdf"):
            got = getattr(b, mode)(di, do).cpu().numpy()
            got = got.T if got.ndim == 2 else got
            worst = max(worst, check_contract(f"abc/{name}/{mode}", got, oracle.eval(ob, i, o, None, mode)))
    print(f"\ncontract abc vs oracle: worst relative difference {worst:.3e}")
    assert worst > 0.0, "the value-contract kernels did not run (results are bit-identical to the oracle)"
    # outside the domain: bit-exact
    row = np.array(param_tables.abc_params("gold-metallic-paint"), np.float64)
    djb.set_contract_1e5(ct_ctx, False)
    for what in ("ior", "exponent"):
        r2 = row.copy()
        if what == "ior": r2[8] = 0.9
        else: r2[7] = 40.0
        mk = lambda r2=r2: djb.abc.from_params(r2, ctx=ct_ctx)
        want = mk().eval(di, do)
        djb.set_contract_1e5(ct_ctx, True)
        got = mk().eval(di, do)
        djb.set_contract_1e5(ct_ctx, False)
        same = (got.view(__import__("torch").int32) == want.view(__import__("torch").int32)) | (got.isnan() & want.isnan())
        assert bool(same.all()), f"contract mode changed a result outside its domain ({what})"
    djb.set_contract_1e5(ct_ctx, True)


# ---------------------------------------------------------------------------------------------- SGD (sgd::eval)
def test_contract_sgd_all_materials_selftest(gpu_ctx):
    """The SGD fast path (k_ct_fast_v4<SGD>) against the bit-exact per-pair code on the device: the 100 published rows x the
    five input families, 2^22 generated pairs each (k up to 856, c up to 1e38, lambda up to 1.5e7, alpha down to 1.6e-5).
    Its shadowing term is a wall: which share of the pairs the error bound hands to tier 2 depends on the material -- printed
    per family, and reported per leg by bench.py (secondary.*_contract)."""
    worst = 0.0
    shares = np.zeros((100, 5))
    for k, name in enumerate(synth.MERL_NAMES):
        b = djb.sgd(name, ctx=gpu_ctx)
        for family in range(5):
            r = djb.selftest_contract(b, None, n=1 << 22, seed=900 + 5 * k + family, family=family, ctx=gpu_ctx)
            assert r["pairs"] == 1 << 22
            assert r["zero_mismatch"] == 0 and r["outside_1e5"] == 0, (name, family, r)
            worst = max(worst, r["max_rel_eval"], r["max_rel_pdf"])
            shares[k, family] = r["tier2"] / r["pairs"]
    s0 = np.sort(shares[:, 0])
    print(f"\ncontract sgd: worst relative difference over 100 materials x 5 families {worst:.3e}; tier-2 share on the bench "
          f"distribution: median {s0[50]:.3f}, 90th percentile {s0[90]:.3f}, max {s0[-1]:.3f}; {int((s0 < 0.3).sum())} materials below 0.3")
    assert worst <= RTOL
    assert (s0 < 0.3).sum() >= 50, "the sgd fast path covers too few materials to be worth its kernel"


def test_contract_sgd_vs_oracle(ct_ctx, oracle):
    """eval / evalp / pdf through the batch API (dense device views -> the two-tier kernels) against the oracle, hostile
    inputs included; an sgd object whose Fresnel term was replaced stays on the bit-exact kernel."""
    rng = np.random.default_rng(6)
    n = (1 << 16) + 1
    i = synth.directions_aos(n, 31).copy(); o = synth.directions_aos(n, 32).copy()
    i[:2000, 2] *= -1; o[2000:4000, 2] *= -1
    i[4000:4100] = np.nan; o[4100:4200, 0] = np.inf
    i[4200:6200] = o[4200:6200] * np.float32([-1, -1, 1]) + rng.normal(0, 1e-4, (2000, 3)).astype(np.float32)
    o[6200:8200] = i[6200:8200]
    i[8200:9200] *= np.float32(3.0)
    i[9200:11200, 2] *= np.float32(1e-3); i[9200:11200] /= np.linalg.norm(i[9200:11200], axis=1, keepdims=True)   # grazing: up the wall
    i = i.astype(np.float32); o = o.astype(np.float32)
    di, do = soa(i), soa(o)
    worst, differs = 0.0, 0
    for name in ("gold-metallic-paint", "alum-bronze", "alumina-oxide", "black-fabric", "chrome", "white-marble", "yellow-plastic", "teflon"):
        b, ob = djb.sgd(name, ctx=ct_ctx), oracle.sgd(name)
        for mode in ("eval", "evalp", "pdf"):
            got = getattr(b, mode)(di, do).cpu().numpy()
            got = got.T if got.ndim == 2 else got
            want = oracle.eval(ob, i, o, None, mode)
            worst = max(worst, check_contract(f"sgd/{name}/{mode}", got, want))
            differs += int(np.sum(np.ascontiguousarray(got, np.float32).view(np.uint32) != want.view(np.uint32)))
    print(f"\ncontract sgd vs oracle: worst relative difference {worst:.3e}")
    assert differs > 0, "the value-contract kernels did not run (results are bit-identical to the oracle)"


# ---------------------------------------------------------------- Beckmann `sample` under the contract (round 4)
SAMPLE_PARAMS = [None, ("elliptic", 0.2, 0.5, 0.7), ("elliptic", 0.3, 0.3, 0.0), ("elliptic", 0.02, 0.02, 0.0),
                 ("pdfparams", 0.4, 0.25, 0.6, 0.1, -0.2), ("elliptic", 0.05, 0.8, 0.3)]
ATOL_DIR = 1e-5          # every component of the sampled unit vector (x |o| for an un-normalised view direction)


def check_directions(name, got, want, o):
    got = np.asarray(got, np.float64); want = np.asarray(want, np.float64)
    assert np.array_equal(np.isnan(got), np.isnan(want)), f"{name}: NaN pattern differs"
    with np.errstate(invalid="ignore", over="ignore"):
        on = np.linalg.norm(np.asarray(o, np.float64), axis=1, keepdims=True)
    tol = ATOL_DIR * np.where(np.isfinite(on), np.maximum(1.0, on), 1.0)
    m = np.isfinite(want) & np.isfinite(got)
    assert np.array_equal(np.isfinite(got), np.isfinite(want)), f"{name}: Inf pattern differs"
    with np.errstate(invalid="ignore"):
        d = np.where(m, np.abs(got - want), 0.0)
    assert np.all(d <= tol), f"{name}: {int(np.sum(d > tol))} components outside the contract, worst {float(d.max()):.3e}"
    # the reference's degenerate answer (0, 0, 1) (stretched view direction below the horizon) is a decision, not a value
    deg = np.all(want == np.array([0.0, 0.0, 1.0]), axis=1)
    assert np.array_equal(got[deg], want[deg]), f"{name}: degenerate samples differ"
    return float(d.max())


@pytest.mark.parametrize("ndf", ["beckmann", "ggx"])
def test_contract_sample_vs_oracle(ct_ctx, oracle, ndf):
    """djb_sample_batch / djb_sample_rng_batch with DJB_OPT_CONTRACT_1E5 (Beckmann: the fp32 Newton sequence; GGX: the closed forms):
    directions within 1e-5 per component of the oracle's, on the bench inputs, a grazing and a near-normal family and un-normalised
    view directions; evalp_is keeps the reference's direction bit for bit"""
    n = 1 << 18
    u1, u2 = synth.uniforms(n, synth.SEED_U1), synth.uniforms(n, synth.SEED_U2)
    base = synth.directions_aos(n, synth.SEED_O)
    graz = base.copy(); graz[:, 2] = 0.02 + 0.05 * graz[:, 2]; graz /= np.linalg.norm(graz, axis=1, keepdims=True)
    near = base * np.array([0.01, 0.01, 0.0], np.float32) + np.array([0, 0, 1], np.float32); near /= np.linalg.norm(near, axis=1, keepdims=True)
    long_ = base * (0.25 + 3.0 * synth.uniforms(n, 77))[:, None]
    import torch
    b = getattr(djb, ndf)(ctx=ct_ctx); ob = oracle.microfacet(ndf)
    worst, differs = 0.0, 0
    for fam, o in (("bench", base), ("grazing", graz.astype(np.float32)), ("near-normal", near.astype(np.float32)), ("un-normalised", long_.astype(np.float32))):
        do, d1, d2 = soa(o), torch.from_numpy(u1).cuda(), torch.from_numpy(u2).cuda()
        for p in SAMPLE_PARAMS:
            got = b.sample(d1, d2, do, mk_params(p)).cpu().numpy().T
            want = oracle.sample(ob, u1, u2, o, p)
            worst = max(worst, check_directions(f"sample/{fam}/{p}", got, want, o))
            differs += int(np.sum(np.ascontiguousarray(got, np.float32).view(np.uint32) != want.view(np.uint32)))
    assert differs > 0, "contract mode returned bit-identical directions everywhere: the fast path did not run"
    # evalp_is keeps the reference's DIRECTION under the option (its pdf moves by 1e-3 for a 1e-5 change of direction)
    o = base[:4096]; p = ("elliptic", 0.2, 0.5, 0.7)
    w, i_, pdf = b.evalp_is(torch.from_numpy(u1[:4096]).cuda(), torch.from_numpy(u2[:4096]).cuda(), soa(o), mk_params(p))
    ww, wi, wpdf = oracle.evalp_is(ob, u1[:4096], u2[:4096], o, p)
    bits = lambda a: np.ascontiguousarray(a, np.float32).view(np.uint32)
    assert np.array_equal(bits(i_.cpu().numpy().T), bits(wi))
    print(f"\ncontract-mode {ndf} sample: worst component difference vs the oracle {worst:.3e} (contract {ATOL_DIR})")


@pytest.mark.parametrize("ndf", ["beckmann", "ggx"])
def test_contract_sample_selftest(gpu_ctx, ndf):
    """djb_selftest_contract_sample: 2^26 generated samples per lobe and family against the bit-exact per-sample code on the
    device -- nothing the fast path keeps may be outside 1e-5, and its per-sample error bound must bound (usage < 1)"""
    b = getattr(djb, ndf)(ctx=gpu_ctx)
    for p in SAMPLE_PARAMS[1:] + [("elliptic", 1.0, 1.0, 0.0)]:
        for family in range(5):
            r = djb.selftest_contract_sample(b, mk_params(p), n=1 << 26, seed=21 + family, family=family, ctx=gpu_ctx)
            assert r["outside_1e5"] == 0 and r["max_abs_dir"] <= ATOL_DIR, (p, family, r)
            assert r["bound_used"] < 1.0, (p, family, r)
    r = djb.selftest_contract_sample(b, mk_params(("elliptic", 0.2, 0.5, 0.7)), n=1 << 26, seed=5, family=0, ctx=gpu_ctx)
    assert r["exact_path"] < 0.15 * r["samples"], f"the fast path keeps too little of the bench distribution: {r}"
    with pytest.raises(djb.exc):          # outside the sampler's domain (roughness below 1e-3)
        djb.selftest_contract_sample(b, mk_params(("elliptic", 1e-4, 0.3, 0.0)), n=1024, ctx=gpu_ctx)


@pytest.mark.parametrize("ndf", ["ggx", "beckmann"])
def test_contract_evalp_is_vs_oracle(ct_ctx, oracle, ndf):
    """evalp_is under DJB_OPT_CONTRACT_1E5: the sampled direction is the reference's, bit for bit; weight and pdf are within
    1e-5 relative with the reference's zero pattern (ct_is_tail) -- bench inputs, grazing and near-normal views, three
    Fresnel terms, shadowing on / off, dense device batches (the fast path) and a host batch"""
    import torch
    n = 1 << 17
    u1, u2 = synth.uniforms(n, synth.SEED_U1), synth.uniforms(n, synth.SEED_U2)
    base = synth.directions_aos(n, synth.SEED_O)
    graz = base.copy(); graz[:, 2] = 0.02 + 0.05 * graz[:, 2]; graz /= np.linalg.norm(graz, axis=1, keepdims=True)
    near = base * np.array([0.01, 0.01, 0.0], np.float32) + np.array([0, 0, 1], np.float32); near /= np.linalg.norm(near, axis=1, keepdims=True)
    below = base.copy(); below[::7, 2] *= -1                       # some views below the horizon: weight, pdf and direction all zero
    bits = lambda a: np.ascontiguousarray(a, np.float32).view(np.uint32)
    worst, differs = 0.0, 0
    d1, d2 = torch.from_numpy(u1).cuda(), torch.from_numpy(u2).cuda()
    for fres in (("ideal",), ("schlick", 1.0, 0.71, 0.29), ("unpolarized", 1.5, 1.8, 2.4)):
        for shadow in (True, False):
            g = getattr(djb, ndf)(mk_fresnel(fres), shadow, ctx=ct_ctx)
            ob = oracle.microfacet(ndf, fres, shadow)
            for fam, o in (("bench", base), ("grazing", graz.astype(np.float32)), ("near-normal", near.astype(np.float32)), ("below", below)):
                for p in PARAMS[:4] if fam == "bench" else PARAMS[2:3]:
                    w, i_, pdf = g.evalp_is(d1, d2, soa(o), mk_params(p))
                    ww, wi, wpdf = oracle.evalp_is(ob, u1, u2, o, p)
                    name = f"{ndf}/{fres[0]}/{shadow}/{fam}/{p}"
                    assert np.array_equal(bits(i_.cpu().numpy().T), bits(wi)), f"{name}: sampled directions are not the reference's"
                    worst = max(worst, check_contract(name + "/weight", w.cpu().numpy().T, ww))
                    worst = max(worst, check_contract(name + "/pdf", pdf.cpu().numpy(), wpdf))
                    differs += int(np.sum(bits(pdf.cpu().numpy()) != bits(wpdf)))
    assert differs > 0, "contract mode returned bit-identical pdfs everywhere: the fast tail did not run"
    # a host batch through the same option
    g = getattr(djb, ndf)(ctx=ct_ctx); ob = oracle.microfacet(ndf)
    p = ("elliptic", 0.2, 0.5, 0.7)
    w, i_, pdf = g.evalp_is(u1[:5000], u2[:5000], base[:5000], mk_params(p))
    ww, wi, wpdf = oracle.evalp_is(ob, u1[:5000], u2[:5000], base[:5000], p)
    assert np.array_equal(bits(i_), bits(wi))
    check_contract("host/weight", w, ww); check_contract("host/pdf", pdf, wpdf)
    print(f"\ncontract-mode {ndf} evalp_is: max relative error of weight / pdf vs the oracle {worst:.3e} (contract {RTOL})")


@pytest.mark.parametrize("ndf", ["beckmann", "ggx"])
def test_contract_sample_hostile_inputs(gpu_ctx, ndf):
    """NaN / Inf / zero / un-normalised / below-horizon view directions, uniforms outside [0, 1) and NaN, a ragged batch size,
    a strided (array-of-vec3) layout: under the option the sampler returns the exact kernel's NaNs and degenerate answers,
    and stays within the contract everywhere else"""
    import torch
    n = (1 << 16) + 5
    o = synth.directions_aos(n, 31).copy()
    u1, u2 = synth.uniforms(n, 32).copy(), synth.uniforms(n, 33).copy()
    k = n // 10
    o[:k, 2] *= -1                                                   # below the horizon
    o[k:2 * k] *= 1e-3; o[2 * k:3 * k] *= 250.0                     # un-normalised, both ways
    o[3 * k:3 * k + 50] = 0.0
    o[3 * k + 50:3 * k + 100, 0] = np.nan; o[3 * k + 100:3 * k + 150, 2] = np.inf
    o[4 * k:5 * k, 2] = 1e-4 * np.abs(o[4 * k:5 * k, 2])             # grazing
    o[5 * k:6 * k, :2] *= 1e-5                                       # on the normal
    u1[6 * k:6 * k + 100] = np.nan; u2[6 * k + 100:6 * k + 200] = np.nan
    u1[6 * k + 200:6 * k + 300] = -0.5; u2[6 * k + 300:6 * k + 400] = 1.5; u1[6 * k + 400:6 * k + 500] = 1.0; u2[6 * k + 500:6 * k + 600] = 0.0
    u1[7 * k:8 * k] *= 1e-4; u2[8 * k:9 * k] = 1.0 - 1e-4 * u2[8 * k:9 * k]    # the tails of both uniforms
    b = getattr(djb, ndf)(ctx=gpu_ctx)
    d1, d2 = torch.from_numpy(u1).cuda(), torch.from_numpy(u2).cuda()
    for p in (("elliptic", 0.2, 0.5, 0.7), ("pdfparams", 0.4, 0.25, 0.6, 0.1, -0.2), None):
        for lay in ("soa", "aos"):
            do = soa(o) if lay == "soa" else torch.from_numpy(o).cuda()
            exact = b.sample(d1, d2, do, mk_params(p)).cpu().numpy()
            djb.set_contract_1e5(gpu_ctx, True)
            try:
                got = b.sample(d1, d2, do, mk_params(p)).cpu().numpy()
            finally:
                djb.set_contract_1e5(gpu_ctx, False)
            if lay == "soa":
                exact, got = exact.T, got.T
            check_directions(f"hostile/{lay}/{p}", got, exact, o)


@pytest.mark.parametrize("ndf", ["beckmann", "ggx"])
def test_contract_sample_attack(gpu_ctx, ndf):
    """directed search instead of sampling (tools/contract_sample_attack.py, shorter): candidates hill-climb over the bit patterns
    of (u1, u2, o) to maximise the contract-vs-exact difference; nothing the fast path keeps may leave the contract"""
    import torch
    m = 1 << 15
    b = getattr(djb, ndf)(ctx=gpu_ctx)
    for p in (("elliptic", 0.2, 0.5, 0.7), ("elliptic", 1.0, 1.0, 0.0)):
        o = djb.gen_directions(m, 61, ctx=gpu_ctx)
        u1, u2 = djb.gen_uniforms(m, 62, ctx=gpu_ctx), djb.gen_uniforms(m, 63, ctx=gpu_ctx)
        best, c = djb.contract_sample_attack(b, u1, u2, o, mk_params(p), iters=256, seed=9, ctx=gpu_ctx)
        assert c["outside"] == 0 and float(best.max()) < 1.0, (p, c, float(best.max()))
        assert c["accepted"] > 0 and c["evaluations"] >= m * 200


# ---------------------------------------------------------------- utia::eval under the contract (round 6)
def hostile_pairs(n, seed):
    rng = np.random.default_rng(seed)
    i = synth.directions_aos(n, 11).copy(); o = synth.directions_aos(n, 12).copy()
    k = n // 8
    o[:k, 2] *= -1                                            # below the horizon
    i[k:2 * k, 2] *= 1e-4; i[k:2 * k] /= np.linalg.norm(i[k:2 * k], axis=1, keepdims=True)       # grazing
    o[2 * k:3 * k] = i[2 * k:3 * k] * np.array([-1, -1, 1], np.float32)                             # mirror pairs
    o[3 * k:4 * k] = i[3 * k:4 * k]                                                                  # identical
    # un-normalised, shorter than 1: |z| > 1, a NaN or an infinite component gives a NaN angle, and the reference then indexes its table
    # out of bounds -- no defined answer to hold anything against (tools/hostile_parity_sweep.py leaves those out for utia as well)
    i[4 * k:5 * k] *= rng.uniform(0.1, 1.0, (k, 1)).astype(np.float32)
    o[5 * k:5 * k + 8] = 0
    o[5 * k + 24:5 * k + 32, 2] = 1e-20
    i[6 * k:7 * k, :2] *= 1e-3; i[6 * k:7 * k] /= np.linalg.norm(i[6 * k:7 * k], axis=1, keepdims=True)   # near-normal
    # azimuths on the cell boundaries of the 7.5-degree grid and polar angles on those of the 15-degree grid
    m = k // 2
    phi = np.deg2rad(7.5 * rng.integers(0, 48, m)); th = np.deg2rad(rng.uniform(1, 89, m))
    o[7 * k:7 * k + m] = np.stack([np.sin(th) * np.cos(phi), np.sin(th) * np.sin(phi), np.cos(th)], 1)
    th = np.deg2rad(15.0 * rng.integers(0, 7, m)); phi = rng.uniform(0, 2 * np.pi, m)
    i[7 * k:7 * k + m] = np.stack([np.sin(th) * np.cos(phi), np.sin(th) * np.sin(phi), np.cos(th)], 1)
    return i.astype(np.float32), o.astype(np.float32)


def test_contract_utia_vs_oracle(ct_ctx, oracle, tmp_path):
    """DJB_OPT_CONTRACT_1E5 on utia::eval: cells, weights and the 16-tap sums remain the reference's bits; the sRGB power runs on the
    fast transcendentals.  Against the oracle: zeros / NaNs where it has them, everything else within 1e-5 relative -- on the bench
    pairs and on hostile ones, a rough and a smooth table (the latter with negative samples: clamped at load)."""
    n = N + 3
    for name, tab in (("uniform", np.random.default_rng(11).uniform(-5.0, 120.0, size=3 * 288 * 288)), ("smooth", synth.utia_table_smooth())):
        p = str(tmp_path / f"{name}.bin"); np.asarray(tab, np.float64).tofile(p)
        u, ou = djb.utia(p, ctx=ct_ctx), oracle.utia(p)
        for fam, (i, o) in (("bench", (synth.directions_aos(n, synth.SEED_I), synth.directions_aos(n, synth.SEED_O))), ("hostile", hostile_pairs(n, 5))):
            with np.errstate(all="ignore"):
                for op in ("eval", "evalp"):
                    got = getattr(u, op)(soa(i), soa(o)).cpu().numpy().T
                    mx = check_contract(f"utia {name} {fam} {op}", got, oracle.eval(ou, i, o, None, op))
                    print(f"utia contract {name:8s} {fam:8s} {op:6s} max rel err {mx:.3e}")
                fr, pdf = u.eval_pdf(soa(i), soa(o))
                check_contract(f"utia {name} {fam} eval+pdf", fr.cpu().numpy().T, oracle.eval(ou, i, o, None, "eval"))
                assert np.array_equal(pdf.cpu().numpy().view(np.uint32), oracle.eval(ou, i, o, None, "pdf").view(np.uint32))


def test_contract_utia_full_size_against_the_exact_kernels(gpu_ctx):
    """5e7 bench pairs: the contract launch against the bit-exact launch -- same zeros, max relative difference inside the contract;
    strided (array-of-vec3) views as well."""
    import torch
    n = 50_000_000
    i = djb.gen_directions(n, synth.SEED_I, ctx=gpu_ctx); o = djb.gen_directions(n, synth.SEED_O, ctx=gpu_ctx)
    u = djb.utia.from_table(np.random.default_rng(11).uniform(0.0, 120.0, size=3 * 288 * 288), ctx=gpu_ctx)
    exact = u.eval(i, o)
    try:
        djb.set_contract_1e5(gpu_ctx, True)
        fast = u.eval(i, o)
        m = 1_000_001
        ia = i[:, :m].t().contiguous(); oa = o[:, :m].t().contiguous()
        fast_aos = u.eval(ia, oa)
    finally:
        djb.set_contract_1e5(gpu_ctx, False)
    assert torch.equal(exact == 0, fast == 0)
    rel = ((fast - exact).abs() / exact.abs().clamp_min(1e-30)).masked_fill(exact == 0, 0.0)
    mx = float(rel.max())
    print(f"utia contract vs exact, {n} pairs: max rel {mx:.3e}, identical bits {float((fast.view(torch.int32) == exact.view(torch.int32)).float().mean()):.4f}")
    assert mx <= RTOL
    assert float(exact.abs().sum()) > 0
    assert torch.equal(fast_aos.t().contiguous().view(torch.int32), fast[:, :m].contiguous().view(torch.int32))
