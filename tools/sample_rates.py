#!/usr/bin/env python3
"""sample() / evalp_is() throughput of the two analytic lobes on 2e8 device-resident directions with
on-chip uniforms (run on the GPU box).  PYTHONPATH=. python tools/sample_rates.py"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dj_brdf_amd import djb, synth, _lib
ctx = djb.default_context(0); lib = _lib.load()
n = 200_000_000
o = djb.gen_directions(n, synth.SEED_O, ctx=ctx)
u1 = djb.gen_uniforms(n, synth.SEED_U1, ctx=ctx); u2 = djb.gen_uniforms(n, synth.SEED_U2, ctx=ctx)
out_i = torch.empty((3, n), dtype=torch.float32, device=o.device); out_w = torch.empty_like(out_i)
pdf = torch.empty((n,), dtype=torch.float32, device=o.device)
vo, vi, vw = djb._Vec(o), djb._Vec(out_i), djb._Vec(out_w)
p = djb.microfacet.params.elliptic(0.2, 0.5, 0.7)
lobes = [("ggx", djb.ggx(ctx=ctx)), ("beckmann", djb.beckmann(ctx=ctx))]
if os.environ.get("DJB_SAMPLE_RATES_TABULAR"):      # the fitted lobe the dj_merl / dj_utia plugins sample at render time (nmap scheme)
    lobes.append(("tabular", djb.tabular(djb.ggx(ctx=ctx), 90, True, ctx=ctx)))
if os.environ.get("DJB_SAMPLE_RATES_CONTRACT"):     # DJB_OPT_CONTRACT_1E5: Beckmann sample within 1e-5, evalp_is weights / pdfs within 1e-5 of the exact direction's
    djb.set_contract_1e5(ctx, True); print("# DJB_OPT_CONTRACT_1E5 on")
for name, b in lobes:
    def sample():
        _lib.check(lib.djb_sample_batch(ctx._h, b._h, C.c_int64(n), C.c_void_p(u1.data_ptr()), C.c_void_p(u2.data_ptr()),
                                        C.byref(vo.view), C.byref(p._p), C.byref(vi.view), C.c_int(0)))
    def evalp_is():
        _lib.check(lib.djb_evalp_is_batch(ctx._h, b._h, C.c_int64(n), C.c_void_p(u1.data_ptr()), C.c_void_p(u2.data_ptr()),
                                          C.byref(vo.view), C.byref(p._p), C.byref(vw.view), C.byref(vi.view),
                                          C.c_void_p(pdf.data_ptr()), C.c_int(0)))
    for tag, f, bytes_ in (("sample", sample, 32), ("evalp_is", evalp_is, 48)):
        for _ in range(6): f()      # steady clocks
        torch.cuda.synchronize(); ctx.timer_start()
        for _ in range(6): f()
        ms = ctx.timer_stop_ms() / 6
        print(f"{name:9s} {tag:9s}: {ms:7.3f} ms per 2e8 -> {n/ms/1e6:6.1f} G/s  ({bytes_*n/ms/1e6/8000*100:4.1f} % of HBM at {bytes_} B/unit)")
