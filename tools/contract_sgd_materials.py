#!/usr/bin/env python3
"""DJB_OPT_CONTRACT_1E5 for sgd::eval, material by material (run on the GPU box): tier-2 share and measured error of the
fast path on the bench distribution (djb_selftest_contract, 2^22 pairs), then the eval rate of a few materials with the
option off / on, 1e8 device-resident pairs.  PYTHONPATH=. python tools/contract_sgd_materials.py [n_rate_materials]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dj_brdf_amd import djb, synth, _lib

ctx = djb.default_context(0); lib = _lib.load()
rows = []
for name in synth.MERL_NAMES:
    b = djb.sgd(name, ctx=ctx)
    r = djb.selftest_contract(b, None, n=1 << 22, seed=5, family=0, ctx=ctx)
    rows.append((name, r["tier2"] / r["pairs"], max(r["max_rel_eval"], r["max_rel_pdf"]), r["outside_1e5"], r["zero_mismatch"]))
sh = np.array([r[1] for r in rows])
print(f"# sgd contract mode, bench distribution, 2^22 pairs per material: tier-2 share median {np.median(sh):.4f}, mean {sh.mean():.4f}, "
      f"90th pct {np.sort(sh)[90]:.4f}, max {sh.max():.4f}; materials below 0.05: {(sh < 0.05).sum()}, below 0.3: {(sh < 0.3).sum()}; "
      f"worst error {max(r[2] for r in rows):.3e}; outside 1e-5: {sum(r[3] for r in rows)}; zero-pattern mismatches: {sum(r[4] for r in rows)}")
for name, s, e, out, z in rows:
    print(f"{name:28s} tier2 {s:8.5f}  max rel err {e:.3e}  outside {out}  zero {z}")
n = 100_000_000
i = djb.gen_directions(n, synth.SEED_I, ctx=ctx); o = djb.gen_directions(n, synth.SEED_O, ctx=ctx)
out = torch.empty((3, n), dtype=torch.float32, device=i.device)
vi, vo, vout = djb._Vec(i), djb._Vec(o), djb._Vec(out)
order = np.argsort(sh)
pick = ["gold-metallic-paint"] + [rows[k][0] for k in (order[0], order[25], order[50], order[75], order[99])]
print("# rates, 1e8 pairs: exact kernel / contract mode")
for name in pick[: int(sys.argv[1]) if len(sys.argv) > 1 else 6]:
    b = djb.sgd(name, ctx=ctx)
    res = []
    for on in (False, True):
        djb.set_contract_1e5(ctx, on)
        for _ in range(4):
            _lib.check(lib.djb_eval_batch(ctx._h, b._h, C.c_int64(n), C.byref(vi.view), C.byref(vo.view), None, C.byref(vout.view), C.c_int(0)))
        torch.cuda.synchronize(); ctx.timer_start()
        for _ in range(5):
            _lib.check(lib.djb_eval_batch(ctx._h, b._h, C.c_int64(n), C.byref(vi.view), C.byref(vo.view), None, C.byref(vout.view), C.c_int(0)))
        res.append(ctx.timer_stop_ms() / 5)
    djb.set_contract_1e5(ctx, False)
    s = sh[[r[0] for r in rows].index(name)]
    print(f"{name:28s} tier2 {s:7.4f}: {res[0]:7.3f} ms -> {res[1]:7.3f} ms per 1e8 ({36 * n / res[1] / 1e6 / 8000 * 100:5.1f} % of HBM at 36 B/pair)")
