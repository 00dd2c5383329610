#!/usr/bin/env python3
"""utia::eval throughput of the library at DJB_LIB_PATH (1e8 device-resident pairs; run on the GPU box)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dj_brdf_amd import djb, synth, _lib
ctx = djb.default_context(0); lib = _lib.load()
n = 100_000_000
i = djb.gen_directions(n, synth.SEED_I, ctx=ctx); o = djb.gen_directions(n, synth.SEED_O, ctx=ctx)
u = djb.utia.from_table(np.random.default_rng(11).uniform(0.0, 120.0, size=3 * 288 * 288), ctx=ctx)
out = torch.empty((3, n), dtype=torch.float32, device=i.device)
vi, vo, vout = djb._Vec(i), djb._Vec(o), djb._Vec(out)
def run(): _lib.check(lib.djb_eval_batch(ctx._h, u._h, C.c_int64(n), C.byref(vi.view), C.byref(vo.view), None, C.byref(vout.view), C.c_int(0)))
def timed(exact_only, reps=20):
    djb.set_utia_exact_only(ctx, exact_only)
    run(); run(); torch.cuda.synchronize(); ctx.timer_start()
    for _ in range(reps): run()
    return ctx.timer_stop_ms() / reps
for _ in range(12): run()          # clocks up before anything is timed
res = {False: [], True: []}
for rnd in range(3):
    for eo in (True, False):
        res[eo].append(timed(eo))
djb.set_utia_exact_only(ctx, True); run(); one = out.clone()
djb.set_utia_exact_only(ctx, False); run(); torch.cuda.synchronize()
same = bool(torch.equal(one.view(torch.int32), out.view(torch.int32)))
fmt = lambda v: "/".join(f"{x:.3f}" for x in v)
print(f"{os.environ.get('DJB_LIB_PATH', 'default')}: utia eval ms per 1e8, alternating: two-tier {fmt(res[False])} "
      f"({n/min(res[False])/1e6:.2f} G eval/s), one kernel with the exact fall-backs inline {fmt(res[True])}; "
      f"identical bits: {same}; checksum {float(out.double().sum()):.6e}")
