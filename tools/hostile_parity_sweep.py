#!/usr/bin/env python3
"""Every BRDF kind x every operator on pairs a renderer's stray rays produce (either direction below / on the horizon, un-normalised,
opposed, grazing, on the normal, equal, NaN / Inf / zero vectors; uniforms outside [0, 1), NaN): the device kernels against the oracle,
value bits with the signs of zeros (NaN payloads aside).  python tools/hostile_parity_sweep.py   (GPU box; prints one line per case)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import oraclelib  # noqa: E402
from dj_brdf_amd import djb, synth  # noqa: E402
from test_gpu_parity import hostile_pairs, value_bits, mk_params  # noqa: E402

ctx = djb.default_context(0) if "--cpu" not in sys.argv else djb.cpu_context()
O = oraclelib.oracle()
n = 1 << 17
dirs = (synth.directions_aos(n, synth.SEED_I), synth.directions_aos(n, synth.SEED_O), synth.uniforms(n, synth.SEED_U1), synth.uniforms(n, synth.SEED_U2))
i, o = hostile_pairs(dirs)
u1, u2 = dirs[2].copy(), dirs[3].copy()
u1[:64] = np.nan; u2[64:128] = np.nan; u1[128:192] = -0.5; u2[192:256] = 1.5; u1[256:320] = 1.0; u2[320:384] = 0.0; u1[384:448] = 0.0; u2[448:512] = 1.0
if ctx.is_cpu:
    n = 1 << 13
    sel = np.r_[0:512, 8192:8192 + 512, 2 * 8192:2 * 8192 + 512, 3 * 8192:3 * 8192 + 512, 4 * 8192:4 * 8192 + 512, 5 * 8192:5 * 8192 + 512, 6 * 8192:6 * 8192 + 512, 7 * 8192:7 * 8192 + 512,
                8 * 8192:8 * 8192 + 512, 9 * 8192:9 * 8192 + 512, 10 * 8192:10 * 8192 + 512]
    i, o, u1, u2 = i[sel], o[sel], u1[sel], u2[sel]
    dev = lambda a: a
else:
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a.T if a.ndim == 2 else a)).cuda()
di, do, d1, d2 = dev(i), dev(o), dev(u1), dev(u2)
host = lambda t: (t.cpu().numpy().T if t.ndim == 2 else t.cpu().numpy()) if not ctx.is_cpu else np.asarray(t)

tab = synth.merl_table_hashed()
np.random.default_rng(3).uniform(-5.0, 130.0, size=3 * 288 * 288).tofile("/tmp/hostile_sweep_utia.bin")      # raw doubles: the UTIA file format
kinds = [
    ("ggx", djb.ggx(ctx=ctx), O.microfacet("ggx"), [None, ("elliptic", 0.2, 0.5, 0.7), ("pdfparams", 0.4, 0.25, 0.3, 0.1, -0.05)]),
    ("ggx schlick noshadow", djb.ggx(djb.fresnel.schlick((1.0, 0.71, 0.29)), False, ctx=ctx), O.microfacet("ggx", ("schlick", 1.0, 0.71, 0.29), False), [("elliptic", 0.3, 0.3, 0.0)]),
    ("beckmann", djb.beckmann(ctx=ctx), O.microfacet("beckmann"), [None, ("elliptic", 0.2, 0.5, 0.7), ("elliptic", 0.05, 0.05, 0.0), ("pdfparams", 0.4, 0.25, 0.3, 0.1, -0.05)]),
    ("beckmann unpol", djb.beckmann(djb.fresnel.unpolarized((1.5, 1.8, 2.4)), True, ctx=ctx), O.microfacet("beckmann", ("unpolarized", 1.5, 1.8, 2.4), True), [("elliptic", 0.3, 0.3, 0.0)]),
    ("merl", djb.merl.from_table(tab, ctx=ctx), O.merl_from_table(tab), [None]),
    ("utia", djb.utia("/tmp/hostile_sweep_utia.bin", ctx=ctx), O.utia("/tmp/hostile_sweep_utia.bin"), [None]),
    ("lambert", djb.lambert(ctx=ctx), O.lambert(), [None]),
    ("sgd", djb.sgd("gold-metallic-paint", ctx=ctx), O.sgd("gold-metallic-paint"), [None]),
    ("abc", djb.abc("gold-metallic-paint", ctx=ctx), O.abc("gold-metallic-paint"), [None]),
]
tg = djb.tabular(djb.ggx(ctx=ctx), 90, True, ctx=ctx)
kinds.append(("tabular(ggx)", tg, O.tabular(O.microfacet("ggx"), 90, True), [None, ("elliptic", 0.3, 0.3, 0.0)]))
kinds.append(("tabular_aniso(beckmann)", djb.tabular_anisotropic(djb.beckmann(ctx=ctx), 12, 16, True, ctx=ctx), O.tabular_anisotropic(O.microfacet("beckmann"), 12, 16, True), [None]))
bad = 0
# utia::eval indexes its table with (int)floor(angle / step) of acos / atan2 results: a NaN, infinite or un-normalised direction (|z| > 1) is
# undefined behaviour in the reference (and a crash in the oracle); its hostile set is the same pairs normalised, the unusable ones dropped
def unit_or_up(v):
    with np.errstate(all="ignore"):
        l = np.linalg.norm(v.astype(np.float64), axis=1, keepdims=True)
        w = (v / l).astype(np.float32)
    good = np.isfinite(w).all(axis=1) & (np.abs(w) <= 1.0).all(axis=1)
    w[~good] = np.array([0, 0, 1], np.float32)
    return w
iu, ou = unit_or_up(i), unit_or_up(o)
diu, dou = dev(iu), dev(ou)
for name, g, ob, plist in kinds:
    if ob is None:
        continue
    if name == "utia":
        for op in ("eval", "evalp", "pdf"):
            got = host(getattr(g, op)(diu, dou)); want = O.eval(ob, iu, ou, None, op)
            m = value_bits(got) != value_bits(want)
            print("%-22s %-40s %-8s %s" % (name, "normalised hostile pairs", op, "ok" if not m.any() else "MISMATCH %d values" % int(m.sum()))); bad += int(m.any())
        continue
    for p in plist:
        up = mk_params(p)
        for op in ("eval", "evalp", "pdf"):
            got = host(getattr(g, op)(di, do, up)); want = O.eval(ob, i, o, p, op)
            m = value_bits(got) != value_bits(want)
            fam = np.unique(np.where(m.reshape(len(i), -1).any(axis=1))[0] // (len(i) // 16 if not ctx.is_cpu else 512))
            print("%-22s %-40s %-8s %s" % (name, p, op, "ok" if not m.any() else "MISMATCH %d values, families %s" % (int(m.sum()), fam.tolist()))); bad += int(m.any())
        got = host(g.sample(d1, d2, do, up)); want = O.sample(ob, u1, u2, o, p)
        m = value_bits(got) != value_bits(want)
        print("%-22s %-40s %-8s %s" % (name, p, "sample", "ok" if not m.any() else "MISMATCH %d values, rows %s" % (int(m.sum()), np.where(m.any(axis=1))[0][:6].tolist()))); bad += int(m.any())
        w, si, pdf = g.evalp_is(d1, d2, do, up); ww, wi, wp = O.evalp_is(ob, u1, u2, o, p)
        m = (value_bits(host(w)) != value_bits(ww)).any(axis=1) | (value_bits(host(si)) != value_bits(wi)).any(axis=1) | (value_bits(host(pdf)) != value_bits(wp))
        print("%-22s %-40s %-8s %s" % (name, p, "evalp_is", "ok" if not m.any() else "MISMATCH %d samples, rows %s" % (int(m.sum()), np.where(m)[0][:6].tolist()))); bad += int(m.any())
# the microfacet queries (ndf, gaf, g1, sigma, vndf, p22, vp22) on the same directions used as h / k / slopes
xy = np.stack([np.tan(3.0 * (u1 - 0.5)), 40.0 * (u2 - 0.5), np.zeros_like(u1)], 1).astype(np.float32)
for name, g, ob, plist in kinds[:4]:
    for p in plist:
        up = mk_params(p)
        for q, dargs, oargs in (("ndf", (di,), (i,)), ("gaf", (di, do, di), (i, o, i)), ("g1", (di, do), (i, o)), ("sigma", (do,), (o,)), ("vndf", (di, do), (i, o)),
                                ("p22", (dev(xy[:, 0]), dev(xy[:, 1])), (xy,)), ("vp22", (dev(xy[:, 0]), dev(xy[:, 1]), do), (xy, o))):
            got = host(getattr(g, q)(*dargs, up)); want = O.microfacet_query(ob, q, *oargs, params=p)
            m = value_bits(got) != value_bits(want)
            print("%-22s %-40s %-8s %s" % (name, p, q, "ok" if not m.any() else "MISMATCH %d values, rows %s" % (int(m.sum()), np.where(m)[0][:6].tolist()))); bad += int(m.any())
# the analytic lobes at the corners of the parameter space (very sharp, very rough, strongly correlated, large mean-normal offsets)
extreme = [("elliptic", 1e-4, 1e-4, 0.0), ("elliptic", 1e-3, 30.0, 1.3), ("elliptic", 50.0, 50.0, 0.0), ("pdfparams", 0.3, 0.3, 0.999, 0.0, 0.0), ("pdfparams", 0.3, 2.0, -0.999, 0.0, 0.0),
           ("pdfparams", 0.5, 0.5, 0.0, 8.0, -8.0), ("pdfparams", 1e-3, 1e-3, 0.0, 0.3, 0.3), ("pdfparams", 20.0, 0.01, 0.9, -2.0, 0.5)]
for name, g, ob, _ in (kinds[0], kinds[2]):
    for p in extreme:
        up = mk_params(p)
        for op in ("eval", "evalp", "pdf"):
            got = host(getattr(g, op)(di, do, up)); want = O.eval(ob, i, o, p, op)
            m = value_bits(got) != value_bits(want)
            print("%-22s %-40s %-8s %s" % (name, p, op, "ok" if not m.any() else "MISMATCH %d values, rows %s" % (int(m.sum()), np.where(m.reshape(len(i), -1).any(axis=1))[0][:6].tolist()))); bad += int(m.any())
        got = host(g.sample(d1, d2, do, up)); want = O.sample(ob, u1, u2, o, p)
        m = value_bits(got) != value_bits(want)
        print("%-22s %-40s %-8s %s" % (name, p, "sample", "ok" if not m.any() else "MISMATCH %d values, rows %s" % (int(m.sum()), np.where(m.any(axis=1))[0][:6].tolist()))); bad += int(m.any())
        w, si, pdf = g.evalp_is(d1, d2, do, up); ww, wi, wp = O.evalp_is(ob, u1, u2, o, p)
        m = (value_bits(host(w)) != value_bits(ww)).any(axis=1) | (value_bits(host(si)) != value_bits(wi)).any(axis=1) | (value_bits(host(pdf)) != value_bits(wp))
        print("%-22s %-40s %-8s %s" % (name, p, "evalp_is", "ok" if not m.any() else "MISMATCH %d samples, rows %s" % (int(m.sum()), np.where(m)[0][:6].tolist()))); bad += int(m.any())
# the per-hit path of dj_beckmannconductor on the same hostile directions: LEAN moments from a bumpy map (plus degenerate records: zero
# variance, negative variance, NaN, huge), every flag combination; per-pair pdfparams records
rg = np.random.default_rng(5)
m = len(i)
lean = np.stack([0.3 * rg.standard_normal(m), 0.3 * rg.standard_normal(m), np.zeros(m), np.zeros(m), 0.02 * rg.standard_normal(m)], 1).astype(np.float32)
lean[:, 2] = lean[:, 0] ** 2 + rg.uniform(1e-4, 0.2, m); lean[:, 3] = lean[:, 1] ** 2 + rg.uniform(1e-4, 0.2, m)
lean[:64, 2] = lean[:64, 0] ** 2; lean[64:128, 3] = 0.5 * lean[64:128, 1] ** 2; lean[128:192, 4] = np.nan; lean[192:256] *= 1e6; lean[256:320] = 0.0
pp = np.stack([rg.uniform(0.02, 1.2, m), rg.uniform(0.02, 1.2, m), rg.uniform(-0.95, 0.95, m), rg.uniform(-0.5, 0.5, m), rg.uniform(-0.5, 0.5, m)], 1).astype(np.float32)
pp[:64, 0] = 0.0; pp[64:128, 2] = 1.0; pp[128:192, 1] = np.nan; pp[192:256, 3] = 50.0
dlean, dpp = (lean, pp) if ctx.is_cpu else (torch.from_numpy(lean).cuda(), torch.from_numpy(pp).cuda())
gb, ogb = djb.beckmann(djb.fresnel.schlick((1.0, 0.71, 0.29)), True, ctx=ctx), O.microfacet("beckmann", ("schlick", 1.0, 0.71, 0.29), True)
base = ("elliptic", 0.1, 0.3, 0.4)
for scale_, filt, biased in ((1.0, True, False), (0.5, False, False), (2.0, True, True)):
    rec = lean.copy()
    if biased: rec[:, :2] += 25.0; rec[:, 4] += 625.0
    drec = rec if ctx.is_cpu else torch.from_numpy(rec).cuda()
    for op in ("eval", "evalp", "pdf"):
        got = host(gb.eval_lean(di, do, mk_params(base), scale_, drec, want=op, filtering=filt, biased=biased)); want = O.eval_lean(ogb, i, o, base, scale_, rec, op, filt, biased)[0]
        mm = value_bits(got) != value_bits(want)
        print("%-22s %-40s %-8s %s" % ("beckmann lean", "scale %g filtering %d biased %d" % (scale_, filt, biased), op, "ok" if not mm.any() else "MISMATCH %d values, rows %s" % (int(mm.sum()), np.where(mm.reshape(m, -1).any(axis=1))[0][:6].tolist()))); bad += int(mm.any())
    w, si, pdf = gb.sample_lean(d1, d2, do, mk_params(base), scale_, drec, evalp_is=True, filtering=filt, biased=biased)
    ww, wi, wp, _ = O.sample_lean(ogb, u1, u2, o, base, scale_, rec, True, filt, biased)
    mm = (value_bits(host(w)) != value_bits(ww)).any(axis=1) | (value_bits(host(si)) != value_bits(wi)).any(axis=1) | (value_bits(host(pdf)) != value_bits(wp))
    print("%-22s %-40s %-8s %s" % ("beckmann lean", "scale %g filtering %d biased %d" % (scale_, filt, biased), "evalp_is", "ok" if not mm.any() else "MISMATCH %d samples, rows %s" % (int(mm.sum()), np.where(mm)[0][:6].tolist()))); bad += int(mm.any())
for name, g, ob in (("ggx pp", kinds[0][1], kinds[0][2]), ("beckmann pp", kinds[2][1], kinds[2][2])):
    for op in ("eval", "evalp", "pdf"):
        got = host(g.eval_pp(di, do, dpp, want=op)); want = O.eval_pp(ob, i, o, pp, op)
        mm = value_bits(got) != value_bits(want)
        print("%-22s %-40s %-8s %s" % (name, "pdfparams records", op, "ok" if not mm.any() else "MISMATCH %d values, rows %s" % (int(mm.sum()), np.where(mm.reshape(m, -1).any(axis=1))[0][:6].tolist()))); bad += int(mm.any())
# brdf::io_to_hd / hd_to_io and the MERL bin index on the same directions
for fn, ofn in (("io_to_hd", O.io_to_hd), ("hd_to_io", O.hd_to_io)):
    ga, gb_ = getattr(djb.brdf, fn)(di, do, ctx=ctx); wa, wb = ofn(i, o)
    mm = (value_bits(host(ga)) != value_bits(wa)) | (value_bits(host(gb_)) != value_bits(wb))
    print("%-22s %-40s %-8s %s" % ("brdf", fn, "", "ok" if not mm.any() else "MISMATCH %d values, rows %s" % (int(mm.sum()), np.where(mm.any(axis=1))[0][:6].tolist()))); bad += int(mm.any())
fin = np.isfinite(i).all(axis=1) & np.isfinite(o).all(axis=1)       # (int) of a NaN angle is undefined in the reference
gi = np.asarray(host(djb.merl_index(di, do, ctx=ctx))) if not ctx.is_cpu else np.asarray(djb.merl_index(i, o, ctx=ctx)); wi_ = O.merl_index(i, o)
mm = (gi.reshape(-1) != wi_.reshape(-1)) & fin
print("%-22s %-40s %-8s %s" % ("merl", "bin index (finite directions)", "", "ok" if not mm.any() else "MISMATCH %d, rows %s" % (int(mm.sum()), np.where(mm)[0][:6].tolist()))); bad += int(mm.any())
print("cases with a mismatch:", bad)
