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
for _ in range(12): run()          # clocks up before anything is timed
torch.cuda.synchronize(); ctx.timer_start()
for _ in range(20): run()
ms = ctx.timer_stop_ms() / 20
print(f"{os.environ.get('DJB_LIB_PATH', 'default')}: utia eval {ms:.3f} ms per 1e8 -> {n/ms/1e6:.2f} G eval/s, checksum {float(out.double().sum()):.6e}")
