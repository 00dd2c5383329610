#!/usr/bin/env python3
"""Beckmann eval + pdf under DJB_OPT_CONTRACT_1E5 by roughness: the share of the bench pairs that tier 1 hands to the exact code
(djb_selftest_contract, 2^26 pairs) and the launch time of 1e8 pairs against the bit-exact kernel.
    PYTHONPATH=. python tools/contract_beckmann_share.py > profiles/r04/contract_beckmann_share.txt      (on the GPU box)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from dj_brdf_amd import djb, synth, _lib  # noqa: E402

ctx = djb.default_context(0); P = djb.microfacet.params; lib = _lib.load()
b = djb.beckmann(ctx=ctx)
n = 100_000_000
i = djb.gen_directions(n, synth.SEED_I, ctx=ctx); o = djb.gen_directions(n, synth.SEED_O, ctx=ctx)
out = torch.empty((3, n), dtype=torch.float32, device=i.device); pdf = torch.empty((n,), dtype=torch.float32, device=i.device)
vi, vo, vout = djb._Vec(i), djb._Vec(o), djb._Vec(out)
print("%-18s %10s %12s %12s %14s %14s" % ("lobe", "tier 2", "max rel eval", "max rel pdf", "contract ms", "exact ms"))
for a in (1.0, 0.3, 0.1, 0.05, 0.02):
    p = P.isotropic(a)
    r = djb.selftest_contract(b, p, n=1 << 26, seed=3, family=0, ctx=ctx)
    assert r["zero_mismatch"] == 0 and r["outside_1e5"] == 0, r
    ms = {}
    for on in (True, False):
        djb.set_contract_1e5(ctx, on)
        def run():
            _lib.check(lib.djb_eval_pdf_batch(ctx._h, b._h, C.c_int64(n), C.byref(vi.view), C.byref(vo.view), C.byref(p._p), C.c_int(0), C.byref(vout.view),
                                              C.c_void_p(pdf.data_ptr()), C.c_int(0)))
        for _ in range(20): run()
        torch.cuda.synchronize(); ctx.timer_start()
        for _ in range(10): run()
        ms[on] = ctx.timer_stop_ms() / 10
    djb.set_contract_1e5(ctx, False)
    print("isotropic(%-5g)   %9.3f%% %12.2e %12.2e %14.3f %14.3f" % (a, 100.0 * r["tier2"] / r["pairs"], r["max_rel_eval"], r["max_rel_pdf"], ms[True], ms[False]))
