"""beckmann::lrep + per-pair params: the batched form of what
dj_beckmannconductor does per hit, against golden vectors from the real reference."""
import os

import numpy as np
import pytest

from dj_brdf_amd import djb
from golden_cases import (LEAN_ANCHOR, LEAN_ANCHOR_E, LEAN_BASE, LEAN_CASES, LEAN_SCALE, PARAM_CASES, lean_texels,
                          lrep_cases)
from test_gpu_parity import assert_close, mk_params

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
OPS = {"add": lambda a, b, x, y: a + b, "mul": lambda a, b, x, y: a * x}


def _lrep(m):
    return djb.beckmann.lrep(*[float(v) for v in m])


def test_lrep_algebra_matches_reference():
    g = np.load(os.path.join(G, "lean.npz"))
    for k, (op, a, b, x, y) in enumerate(lrep_cases()):
        A, B = _lrep(a), _lrep(b)
        if op == "add": R = A + B
        elif op == "mul": R = A * x
        elif op == "iadd": A += B; R = A
        elif op == "imul": A *= x; R = A
        elif op == "shear": A.shear(x, y); R = A
        else: A.scale(x, y); R = A
        got = np.array(djb.beckmann.lrep_to_params(R).get_pdfparams(), np.float32)
        assert np.array_equal(got.view(np.uint32), g[f"lrep{k}"].view(np.uint32)), (k, op, got, g[f"lrep{k}"])
    for k, p in enumerate(PARAM_CASES):
        mp = mk_params(p) or djb.microfacet.params.standard()
        got = np.array(djb.beckmann.lrep_to_params(djb.beckmann.params_to_lrep(mp)).get_pdfparams(), np.float32)
        assert np.array_equal(got.view(np.uint32), g[f"roundtrip{k}"].view(np.uint32)), p
    with pytest.raises(djb.exc):
        djb.beckmann.lrep() * -1.0          # DJB_ASSERT(sc >= 0), dj_brdf.h:2003


@pytest.mark.parametrize("ndf", ["beckmann", "ggx"])
def test_lean_eval_golden(gpu_ctx, ndf):
    g = np.load(os.path.join(G, "lean.npz"))
    b = getattr(djb, ndf)(djb.fresnel.schlick((1.0, 0.71, 0.29)), True, ctx=gpu_ctx)
    base = mk_params(LEAN_BASE)
    # tile the 512 golden hits past the scalar-twin threshold so the kernels (not the host twin) answer
    reps = 4
    i, o = np.tile(g["i"], (reps, 1)), np.tile(g["o"], (reps, 1))
    for c, (scale, filtering, biased) in enumerate(LEAN_CASES):
        tex = np.tile(lean_texels(g["lean"], biased), (reps, 1))
        kw = dict(filtering=filtering, biased=biased)
        wpp = np.tile(g[f"c{c}_pdfparams"], (reps, 1))
        for op in ("eval", "evalp", "pdf"):
            want = np.tile(g[f"c{c}_{ndf}_{op}"], (reps, 1) if op != "pdf" else reps)
            val, pp = b.eval_lean(i, o, base, scale, tex, want=op, return_params=True, **kw)
            assert np.array_equal(pp.view(np.uint32), wpp.view(np.uint32)), f"case {c}: resolved per-pair params differ"
            ex = assert_close(f"{ndf}/lean{c}/{op}", val, want)
            assert ex > 0.9999
            assert_close(f"{ndf}/pp{c}/{op}", b.eval_pp(i, o, wpp, want=op), want)
        fr, pdf = b.eval_lean(i, o, base, scale, tex, want="evalp+pdf", **kw)
        assert_close("fused evalp", fr, np.tile(g[f"c{c}_{ndf}_evalp"], (reps, 1)))
        assert_close("fused pdf", pdf, np.tile(g[f"c{c}_{ndf}_pdf"], reps))


@pytest.mark.parametrize("ndf", ["beckmann", "ggx"])
def test_lean_sample_golden(gpu_ctx, ndf):
    """dj_beckmann_conductor::sample per hit (mitsuba/dj_beckmannconductor.cpp:373-413), batched: per-hit params, then
    evalp_is -- against the real reference's outputs for the same hits."""
    g = np.load(os.path.join(G, "lean.npz"))
    b = getattr(djb, ndf)(djb.fresnel.schlick((1.0, 0.71, 0.29)), True, ctx=gpu_ctx)
    base = mk_params(LEAN_BASE)
    reps = 4
    o, u1, u2 = np.tile(g["o"], (reps, 1)), np.tile(g["u1"], reps), np.tile(g["u2"], reps)
    for c, (scale, filtering, biased) in enumerate(LEAN_CASES):
        tex = np.tile(lean_texels(g["lean"], biased), (reps, 1))
        kw = dict(filtering=filtering, biased=biased)
        w, si, pdf, pp = b.sample_lean(u1, u2, o, base, scale, tex, True, return_params=True, **kw)
        assert np.array_equal(pp.view(np.uint32), np.tile(g[f"c{c}_pdfparams"], (reps, 1)).view(np.uint32))
        assert assert_close(f"{ndf}/lean{c}/is_i", si, np.tile(g[f"c{c}_{ndf}_is_i"], (reps, 1))) > 0.999
        assert_close(f"{ndf}/lean{c}/is_w", w, np.tile(g[f"c{c}_{ndf}_is_w"], (reps, 1)))
        assert_close(f"{ndf}/lean{c}/is_pdf", pdf, np.tile(g[f"c{c}_{ndf}_is_pdf"], reps))
        si2 = b.sample_lean(u1, u2, o, base, scale, tex, False, **kw)
        assert_close(f"{ndf}/lean{c}/sample", si2, np.tile(g[f"c{c}_{ndf}_sample"], (reps, 1)))
        assert np.array_equal(b.sample_pp(u1, u2, o, pp).view(np.uint32), si2.view(np.uint32))
        # scalar-size call (host twin) == kernel
        one = b.sample_lean(u1[:1], u2[:1], o[:1], base, scale, tex[:1], True, **kw)
        for a, k in zip(one, (w, si, pdf)):
            assert np.array_equal(np.asarray(a).view(np.uint32), np.asarray(k[:1]).view(np.uint32))


def test_lean_composition_anchor(gpu_ctx):
    """lrep(lean) * dmapscale + params_to_lrep(base), the plugin's order (mitsuba/dj_beckmannconductor.cpp:296-314):
    the per-pair params the real header gives for one texel at dmapscale 0.7 and 2."""
    b = djb.beckmann(ctx=gpu_ctx)
    n = 4096                                              # device kernel, not the scalar twin
    d = np.tile(np.array([[0.3, 0.2, 0.9327379]], np.float32), (n, 1))
    E = np.tile(np.array([LEAN_ANCHOR_E], np.float32), (n, 1))
    for scale, want in LEAN_ANCHOR.items():
        _, pp = b.eval_lean(d, d, mk_params(LEAN_BASE), scale, E, want="pdf", return_params=True)
        assert np.allclose(pp[0], want, rtol=0, atol=6e-4 if scale == 2.0 else 1e-6), (scale, pp[0])
        assert np.array_equal(pp, np.tile(pp[:1], (n, 1)))
        one = b.eval_lean(d[:1], d[:1], mk_params(LEAN_BASE), scale, E[:1], want="pdf", return_params=True)[1]
        assert np.array_equal(one.view(np.uint32), pp[:1].view(np.uint32)), "scalar twin and kernel disagree"
    with pytest.raises(djb.exc):
        b.eval_lean(d, d, mk_params(LEAN_BASE), -1.0, E, want="pdf")     # DJB_ASSERT(sc >= 0), dj_brdf.h:2024


def test_lean_device_tensors(gpu_ctx):
    import torch
    g = np.load(os.path.join(G, "lean.npz"))
    b = djb.beckmann(ctx=gpu_ctx)
    ti = torch.from_numpy(g["i"].T.copy()).cuda(); to = torch.from_numpy(g["o"].T.copy()).cuda()
    tl = torch.from_numpy(g["lean"]).cuda()
    dev = b.eval_lean(ti, to, mk_params(LEAN_BASE), LEAN_SCALE, tl, want="evalp")
    host = b.eval_lean(g["i"], g["o"], mk_params(LEAN_BASE), LEAN_SCALE, g["lean"], want="evalp")
    assert np.array_equal(dev.cpu().numpy().T.view(np.uint32), host.view(np.uint32))


def test_lean_large_host_batches_are_chunked(gpu_ctx, monkeypatch):
    # the per-pair-parameter host path goes through the same chunked pipeline as eval (both PCIe directions
    # in flight); chunk size forced down, result = the unchunked call bit for bit, resolved params included
    g = np.load(os.path.join(G, "lean.npz"))
    reps = -(-30_011 // len(g["i"]))
    i, o, lean = (np.tile(g[k], (reps, 1))[:30_011] for k in ("i", "o", "lean"))

    def paged(a):        # buffers that own their host pages (inputs and outputs must not share any)
        raw = np.empty(a.size + 3072, np.float32)
        off = (-raw.ctypes.data % 4096) // 4
        v = raw[off:off + a.size].reshape(a.shape); v[...] = a
        return v
    i, o, lean = paged(i), paged(o), paged(lean)
    b = djb.beckmann(djb.fresnel.schlick((1.0, 0.71, 0.29)), True, ctx=gpu_ctx)
    base = mk_params(LEAN_BASE)
    monkeypatch.setenv("DJB_HOST_PIPE_CHUNK", "0")
    want = b.eval_lean(i, o, base, LEAN_SCALE, lean, want="evalp+pdf", return_params=True)
    want_pp = b.eval_pp(i, o, want[2], want="eval")
    monkeypatch.setenv("DJB_HOST_PIPE_CHUNK", "7000")
    monkeypatch.setenv("DJB_HOST_PIPE_REQUIRE", "1")     # the inputs own their pages: falling back would be a bug
    got = b.eval_lean(i, o, base, LEAN_SCALE, lean, want="evalp+pdf", return_params=True)
    for a, w in zip(got, want):
        assert np.array_equal(a.view(np.uint32), w.view(np.uint32))
    assert np.array_equal(b.eval_pp(i, o, paged(want[2]), want="eval").view(np.uint32), want_pp.view(np.uint32))
