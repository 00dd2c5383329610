#!/usr/bin/env python3
"""PCIe-inclusive rate of the DJB_MEM_HOST path: host arrays in, host arrays out, through the C ABI
(djb_eval_batch with mem = DJB_MEM_HOST), caller-owned buffers allocated and touched beforehand as
a C++ caller would.  DESIGN.md section 5 quotes this; it is never the bench `value` (that is
measured with inputs resident in HBM)."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dj_brdf_amd import djb, synth, _lib

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
ctx = djb.default_context(0)
lib = _lib.load()
m = djb.merl.from_table(synth.merl_table(0.3), ctx=ctx)
g = djb.ggx(ctx=ctx)
for layout in ("aos (array of djb::vec3)", "soa"):
    if layout.startswith("aos"):
        i = synth.directions_aos(n, synth.SEED_I); o = synth.directions_aos(n, synth.SEED_O)
        out = np.zeros((n, 3), np.float32)
    else:
        i = np.ascontiguousarray(np.stack(synth.directions(n, synth.SEED_I))); o = np.ascontiguousarray(np.stack(synth.directions(n, synth.SEED_O)))
        out = np.zeros((3, n), np.float32)
    vi, vo, vout = djb._Vec(i), djb._Vec(o), djb._Vec(out)
    for name, b in (("merl.eval", m), ("ggx.eval", g)):
        for chunk in ("0", str(1 << 22), str(1 << 23), str(1 << 24)):
            os.environ["DJB_HOST_PIPE_CHUNK"] = chunk    # 0 = copy in, run, copy out; else chunks with both directions in flight
            best = 1e9
            for _ in range(4):
                t0 = time.perf_counter()
                _lib.check(lib.djb_eval_batch(ctx._h, b._h, C.c_int64(n), C.byref(vi.view), C.byref(vo.view), None,
                                              C.byref(vout.view), C.c_int(_lib.MEM_HOST)))
                best = min(best, time.perf_counter() - t0)
            print(f"{name:10s} host {layout:26s} chunk {int(chunk):>9d} n={n:.0e}  {best*1e3:8.1f} ms  {n/best/1e9:6.3f} G eval/s  "
                  f"({36*n/best/1e9:5.1f} GB/s over PCIe, H2D 24 B + D2H 12 B per eval)", flush=True)

# sample / evalp_is of a Beckmann lobe: 20 B in (u1, u2, o), 12 B (i) or 28 B (i, weight, pdf) out per unit
bk = djb.beckmann(ctx=ctx)
o = synth.directions_aos(n, synth.SEED_O); u1 = synth.uniforms(n, synth.SEED_U1); u2 = synth.uniforms(n, synth.SEED_U2)
oi, ow, opdf = np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32), np.zeros(n, np.float32)
vo, voi, vow = djb._Vec(o), djb._Vec(oi), djb._Vec(ow)
fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
for name, bytes_per, call in (
        ("beckmann.sample", 32, lambda: lib.djb_sample_batch(ctx._h, bk._h, C.c_int64(n), fp(u1), fp(u2), C.byref(vo.view), None,
                                                             C.byref(voi.view), C.c_int(_lib.MEM_HOST))),
        ("beckmann.evalp_is", 48, lambda: lib.djb_evalp_is_batch(ctx._h, bk._h, C.c_int64(n), fp(u1), fp(u2), C.byref(vo.view), None,
                                                                 C.byref(vow.view), C.byref(voi.view), fp(opdf), C.c_int(_lib.MEM_HOST)))):
    for chunk in ("0", str(1 << 23)):
        os.environ["DJB_HOST_PIPE_CHUNK"] = chunk
        best = 1e9
        for _ in range(4):
            t0 = time.perf_counter(); _lib.check(call()); best = min(best, time.perf_counter() - t0)
        print(f"{name:18s} host aos chunk {int(chunk):>9d} n={n:.0e}  {best*1e3:8.1f} ms  {n/best/1e9:6.3f} G/s  "
              f"({bytes_per*n/best/1e9:5.1f} GB/s over PCIe, {bytes_per} B per unit)", flush=True)
