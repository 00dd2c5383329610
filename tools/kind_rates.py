#!/usr/bin/env python3
"""Eval throughput of the BRDF kinds that have no BASELINE config (utia, sgd, abc, tabular), 1e8
device-resident pairs (run on the GPU box).  PYTHONPATH=. python tools/kind_rates.py"""
import sys, time, ctypes as C
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dj_brdf_amd import djb, synth, _lib
ctx = djb.default_context(0); lib = _lib.load()
if os.environ.get("DJB_KIND_RATES_CONTRACT"):      # DJB_OPT_CONTRACT_1E5: the ggx / beckmann ideal and schlick legs run the value-contract kernels
    djb.set_contract_1e5(ctx, True)
    print("# DJB_OPT_CONTRACT_1E5 on (ggx / beckmann with ideal or schlick Fresnel and abc take the two-tier value-contract kernels; the rest is unaffected)")
n = 100_000_000
i = djb.gen_directions(n, synth.SEED_I, ctx=ctx); o = djb.gen_directions(n, synth.SEED_O, ctx=ctx)
rng = np.random.default_rng(11)
tab = rng.uniform(0.0, 120.0, size=3 * 288 * 288)
u = djb.utia.from_table(tab, ctx=ctx)
out = torch.empty((3, n), dtype=torch.float32, device=i.device)
vi, vo, vout = djb._Vec(i), djb._Vec(o), djb._Vec(out)
pdf = torch.empty((n,), dtype=torch.float32, device=i.device)
iso = djb.microfacet.params.isotropic(0.3)
for name, b in (("ggx", djb.ggx(ctx=ctx)), ("beckmann", djb.beckmann(ctx=ctx)), ("beckmann schlick", djb.beckmann(djb.fresnel.schlick((1.0, 0.71, 0.29)), ctx=ctx)),
                ("ggx unpolarized", djb.ggx(djb.fresnel.unpolarized((1.5, 1.8, 2.4)), ctx=ctx)), ("beckmann unpol.", djb.beckmann(djb.fresnel.unpolarized((1.5, 1.8, 2.4)), ctx=ctx))):
    def run():
        _lib.check(lib.djb_eval_pdf_batch(ctx._h, b._h, C.c_int64(n), C.byref(vi.view), C.byref(vo.view), C.byref(iso._p), C.c_int(0), C.byref(vout.view),
                                          C.c_void_p(pdf.data_ptr()), C.c_int(0)))
    for _ in range(10): run()      # steady clocks: these launches take 1-4 ms
    torch.cuda.synchronize(); ctx.timer_start()
    for _ in range(10): run()
    ms = ctx.timer_stop_ms() / 10
    print(f"{name:16s} eval+pdf: {ms:8.3f} ms per 1e8 -> {n/ms/1e6:7.2f} G/s ({40*n/ms/1e6/8000*100:.1f} % of HBM at 40 B/pair)")
for name, b in (("utia", u), ("sgd", djb.sgd("gold-metallic-paint", ctx=ctx)), ("abc", djb.abc("gold-metallic-paint", ctx=ctx)),
                ("tabular(ggx)", djb.tabular(djb.ggx(ctx=ctx), 90, True, ctx=ctx))):
    for _ in range(10):
        _lib.check(lib.djb_eval_batch(ctx._h, b._h, C.c_int64(n), C.byref(vi.view), C.byref(vo.view), None, C.byref(vout.view), C.c_int(0)))
    torch.cuda.synchronize(); ctx.timer_start()
    for _ in range(10):
        _lib.check(lib.djb_eval_batch(ctx._h, b._h, C.c_int64(n), C.byref(vi.view), C.byref(vo.view), None, C.byref(vout.view), C.c_int(0)))
    ms = ctx.timer_stop_ms() / 10
    print(f"{name:14s} eval: {ms:8.3f} ms per 1e8 -> {n/ms/1e6:7.2f} G eval/s ({36*n/ms/1e6/8000*100:.1f} % of HBM)")
