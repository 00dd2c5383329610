#!/usr/bin/env python3
"""DJB_MEM_HOST eval rate against batch size: unchunked (DJB_HOST_PIPE_CHUNK=0) vs the default chunking
(n/8 clamped to [2^19, 2^23] units, batches of >= 2 chunks).  GGX eval, array of djb::vec3 in and out."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dj_brdf_amd import djb, synth, _lib

ctx = djb.default_context(0)
lib = _lib.load()
g = djb.ggx(ctx=ctx)
N = 1 << 25
I = synth.directions_aos(N, synth.SEED_I); O = synth.directions_aos(N, synth.SEED_O); OUT = np.zeros((N, 3), np.float32)
for n in (1 << 18, 1 << 20, 1 << 21, 3 << 20, 1 << 22, 1 << 23, 1 << 24, 1 << 25):
    vi, vo, vout = djb._Vec(I[:n]), djb._Vec(O[:n]), djb._Vec(OUT[:n])
    row = []
    for chunk in ("0", None):
        if chunk is None: os.environ.pop("DJB_HOST_PIPE_CHUNK", None)
        else: os.environ["DJB_HOST_PIPE_CHUNK"] = chunk
        best = 1e9
        for _ in range(6):
            t0 = time.perf_counter()
            _lib.check(lib.djb_eval_batch(ctx._h, g._h, C.c_int64(n), C.byref(vi.view), C.byref(vo.view), None,
                                          C.byref(vout.view), C.c_int(_lib.MEM_HOST)))
            best = min(best, time.perf_counter() - t0)
        row.append(best)
    print(f"n = {n:>9d}   unchunked {row[0]*1e3:8.3f} ms ({n/row[0]/1e9:5.2f} G eval/s)   default {row[1]*1e3:8.3f} ms ({n/row[1]/1e9:5.2f} G eval/s)", flush=True)
