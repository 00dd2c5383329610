#!/usr/bin/env python3
"""Bit-exact Beckmann eval + pdf by roughness: k_eval against k_eval_bk_sharp (pairs whose result is a known zero written at once, the
others evaluated in dense waves).  The kernel is picked per launch by DJB_BK_SHARP_ALPHA (largest alpha that takes the two-path kernel):
    for a in 0 1; do DJB_BK_SHARP_ALPHA=$a PYTHONPATH=. python tools/beckmann_sharp_rates.py; done   (GPU box) -> profiles/r04/beckmann_sharp.txt"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from dj_brdf_amd import djb, synth, _lib  # noqa: E402

ctx = djb.default_context(0); P = djb.microfacet.params; lib = _lib.load()
n = 100_000_000
i = djb.gen_directions(n, synth.SEED_I, ctx=ctx); o = djb.gen_directions(n, synth.SEED_O, ctx=ctx)
out = torch.empty((3, n), dtype=torch.float32, device=i.device); pdf = torch.empty((n,), dtype=torch.float32, device=i.device)
vi, vo, vout = djb._Vec(i), djb._Vec(o), djb._Vec(out)
print("# DJB_BK_SHARP_ALPHA =", os.environ.get("DJB_BK_SHARP_ALPHA", "(shipped default)"))
for name, b in (("ideal", djb.beckmann(ctx=ctx)), ("schlick", djb.beckmann(djb.fresnel.schlick((1.0, 0.71, 0.29)), ctx=ctx))):
    for a in (0.02, 0.05, 0.08, 0.1, 0.12, 0.15, 0.2, 0.3):
        p = P.isotropic(a)
        def run():
            _lib.check(lib.djb_eval_pdf_batch(ctx._h, b._h, C.c_int64(n), C.byref(vi.view), C.byref(vo.view), C.byref(p._p), C.c_int(0), C.byref(vout.view),
                                              C.c_void_p(pdf.data_ptr()), C.c_int(0)))
        for _ in range(10): run()
        torch.cuda.synchronize(); ctx.timer_start()
        for _ in range(10): run()
        ms = ctx.timer_stop_ms() / 10
        zeros = float((out[0] == 0).float().mean())
        print("beckmann %-8s isotropic(%-4g) eval+pdf: %7.3f ms per 1e8 (%.3f of 8 TB/s at 40 B)   zero results %.1f %%" % (name, a, ms, 40 * n / ms / 1e6 / 8000, 100 * zeros))
