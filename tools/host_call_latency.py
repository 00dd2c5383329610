#!/usr/bin/env python3
"""Latency of small DJB_MEM_HOST calls through the C ABI (n = 1 ... 262144), INTEGRATION.md / DESIGN.md section 5."""
import sys, time, ctypes as C
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dj_brdf_amd import djb, _lib
ctx = djb.default_context(0); lib = _lib.load()
g = djb.ggx(ctx=ctx)
for n in (1, 64, 4096, 262144):
    i = np.tile(np.array([[0.3, 0.2, 0.93]], np.float32), (n, 1)); o = np.tile(np.array([[-0.4, 0.1, 0.91]], np.float32), (n, 1))
    out = np.zeros((n, 3), np.float32)
    vi, vo, vout = djb._Vec(i), djb._Vec(o), djb._Vec(out)
    for _ in range(20):
        lib.djb_eval_batch(ctx._h, g._h, C.c_int64(n), C.byref(vi.view), C.byref(vo.view), None, C.byref(vout.view), C.c_int(1))
    t0 = time.perf_counter(); K = 200
    for _ in range(K):
        lib.djb_eval_batch(ctx._h, g._h, C.c_int64(n), C.byref(vi.view), C.byref(vo.view), None, C.byref(vout.view), C.c_int(1))
    dt = (time.perf_counter() - t0) / K
    print(f"host call n={n:7d}: {dt*1e6:8.1f} us per call, {n/dt/1e6:9.2f} M eval/s")
