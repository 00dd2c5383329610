#!/usr/bin/env python3
"""Throughput of the per-pair-parameter path (dj_beckmannconductor's per-hit LEAN code, batched) on 1e8
device-resident pairs: djb_eval_lean_batch evalp+pdf with 5-float LEAN records (run on the GPU box).
PYTHONPATH=. python tools/lean_rates.py"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dj_brdf_amd import djb, synth, _lib
ctx = djb.default_context(0); lib = _lib.load()
n = 100_000_000
i = djb.gen_directions(n, synth.SEED_I, ctx=ctx); o = djb.gen_directions(n, synth.SEED_O, ctx=ctx)
g = torch.Generator(device=i.device); g.manual_seed(7)
lean = torch.empty((n, 5), dtype=torch.float32, device=i.device)
lean[:, 0:2] = (torch.rand((n, 2), generator=g, device=i.device) - 0.5) * 0.2          # E1, E2: mean slopes
lean[:, 2:4] = torch.rand((n, 2), generator=g, device=i.device) * 0.05 + 0.01         # E3, E4: second moments
lean[:, 4] = (torch.rand((n,), generator=g, device=i.device) - 0.5) * 0.01            # E5
out = torch.empty((3, n), dtype=torch.float32, device=i.device); pdf = torch.empty((n,), dtype=torch.float32, device=i.device)
vi, vo, vout = djb._Vec(i), djb._Vec(o), djb._Vec(out)
base = djb.microfacet.params.isotropic(0.1)
for name, b in (("beckmann ideal", djb.beckmann(ctx=ctx)), ("beckmann schlick", djb.beckmann(djb.fresnel.schlick((1.0, 0.71, 0.29)), ctx=ctx)),
                ("ggx ideal", djb.ggx(ctx=ctx))):
    def run():
        _lib.check(lib.djb_eval_lean_batch(ctx._h, b._h, C.c_int64(n), C.byref(vi.view), C.byref(vo.view), C.byref(base._p), C.c_float(1.0), C.c_int(0),
                                           C.c_void_p(lean.data_ptr()), C.c_int(6), C.byref(vout.view), C.c_void_p(pdf.data_ptr()), C.c_void_p(0), C.c_int(0)))
    run(); torch.cuda.synchronize(); ctx.timer_start()
    for _ in range(3): run()
    ms = ctx.timer_stop_ms() / 3
    print(f"{name:17s} eval_lean evalp+pdf: {ms:7.3f} ms per 1e8 -> {n/ms/1e6:6.1f} G/s  ({60*n/ms/1e6/8000*100:4.1f} % of HBM at 60 B/unit)")
