#!/usr/bin/env python3
"""tabular eval (+ pdf) under DJB_OPT_CONTRACT_1E5 by material: the share of generated pairs that tier 1 hands to the exact code and
the largest relative difference among the pairs it keeps (djb_selftest_contract, five input families), and the launch time of 1e8
bench pairs against the bit-exact kernel.
    PYTHONPATH=. python tools/exp/r05/contract_tabular_probe.py > profiles/r05/contract_tabular.txt      (on the GPU box)
Runs against the experiment commit "experiment: contract-mode tier for tabular eval" (reverted: profiles/r05/NOTES.md section 10); on the
shipped library a tabular lobe is outside the contract-mode set and djb_selftest_contract refuses it."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch  # noqa: E402
from dj_brdf_amd import djb, synth, _lib  # noqa: E402

ctx = djb.default_context(0); P = djb.microfacet.params; lib = _lib.load()
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
i = djb.gen_directions(n, synth.SEED_I, ctx=ctx); o = djb.gen_directions(n, synth.SEED_O, ctx=ctx)
out = torch.empty((3, n), dtype=torch.float32, device=i.device); pdf = torch.empty((n,), dtype=torch.float32, device=i.device)
vi, vo, vout = djb._Vec(i), djb._Vec(o), djb._Vec(out)


def sources():
    yield "tabular(ggx, 90)", djb.tabular(djb.ggx(ctx=ctx), 90, True, ctx=ctx), None
    yield "tabular(ggx, 90) elliptic(.2,.5,.7)", djb.tabular(djb.ggx(ctx=ctx), 90, True, ctx=ctx), P.elliptic(0.2, 0.5, 0.7)
    yield "tabular(ggx, 90) isotropic(.1)", djb.tabular(djb.ggx(ctx=ctx), 90, True, ctx=ctx), P.isotropic(0.1)
    yield "tabular(beckmann, 90)", djb.tabular(djb.beckmann(ctx=ctx), 90, True, ctx=ctx), None
    yield "tabular(ggx, 180)", djb.tabular(djb.ggx(ctx=ctx), 180, True, ctx=ctx), None
    for a in (0.3, 0.1, 0.03):
        m = djb.merl.from_table(synth.merl_table(alpha=a), ctx=ctx)
        yield f"tabular(merl ggx-like alpha {a}, 90)", djb.tabular(m, 90, True, ctx=ctx), None
    yield "tabular(merl hashed, 90)", djb.tabular(djb.merl.from_table(synth.merl_table_hashed(), ctx=ctx), 90, True, ctx=ctx), None
    yield "tabular(sgd gold-metallic-paint, 90)", djb.tabular(djb.sgd("gold-metallic-paint", ctx=ctx), 90, True, ctx=ctx), None
    yield "tabular(abc chrome, 90)", djb.tabular(djb.abc("chrome", ctx=ctx), 90, True, ctx=ctx), None


print("%-40s %28s %12s %12s %9s | %21s %21s" % ("lobe", "tier 2 by family 0..4 (%)", "max rel eval", "max rel pdf", "outside", "eval ms ct / exact", "eval+pdf ms ct / exact"))
for name, b, p in sources():
    shares, me, mp, bad = [], 0.0, 0.0, 0
    for family in range(5):
        r = djb.selftest_contract(b, p, n=1 << 24, seed=11 + family, family=family, ctx=ctx)
        shares.append(100.0 * r["tier2"] / r["pairs"]); me = max(me, r["max_rel_eval"]); mp = max(mp, r["max_rel_pdf"])
        bad += r["zero_mismatch"] + r["outside_1e5"]
    ms = {}
    pp = C.byref(p._p) if p is not None else None
    for fused in (False, True):
        for on in (True, False):
            djb.set_contract_1e5(ctx, on)
            def run():
                if fused:
                    _lib.check(lib.djb_eval_pdf_batch(ctx._h, b._h, C.c_int64(n), C.byref(vi.view), C.byref(vo.view), pp, C.c_int(0), C.byref(vout.view),
                                                      C.c_void_p(pdf.data_ptr()), C.c_int(0)))
                else:
                    _lib.check(lib.djb_eval_batch(ctx._h, b._h, C.c_int64(n), C.byref(vi.view), C.byref(vo.view), pp, C.byref(vout.view), C.c_int(0)))
            for _ in range(12): run()
            torch.cuda.synchronize(); ctx.timer_start()
            for _ in range(10): run()
            ms[(fused, on)] = ctx.timer_stop_ms() / 10
    djb.set_contract_1e5(ctx, False)
    print("%-40s %28s %12.2e %12.2e %9d | %9.3f / %9.3f %9.3f / %9.3f" % (name, " ".join("%5.2f" % s for s in shares), me, mp, bad,
          ms[(False, True)], ms[(False, False)], ms[(True, True)], ms[(True, False)]), flush=True)
