#!/usr/bin/env python3
"""Where does a DJB_MEM_HOST call stop being cheaper on the calling thread than on the GPU?  (DJB_SCALAR_HOST_MAX,
include/djb_hip.h).  For n = 16 ... 4096 units per call: the product's host instantiation on one thread (a CPU context
with DJB_CPU_THREADS=1: the code the host twin of a GPU object runs) against the GPU path of the same call (pinned
arena, one launch, one sync: DJB_OPT_SCALAR_ON_DEVICE for n <= the threshold).  -> profiles/r03/scalar_latency.txt"""
import os
import sys
import time

import numpy as np

os.environ.setdefault("DJB_CPU_THREADS", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dj_brdf_amd import djb, synth      # noqa: E402

gpu, cpu = djb.default_context(0), djb.Context("cpu")
djb.set_scalar_on_device(gpu, True)
i_all, o_all = synth.directions_aos(4096, synth.SEED_I), synth.directions_aos(4096, synth.SEED_O)
u1, u2 = synth.uniforms(4096, synth.SEED_U1), synth.uniforms(4096, synth.SEED_U2)
p = djb.microfacet.params.isotropic(0.3)
tab = synth.merl_table_hashed()
cases = []
for ctx in (gpu, cpu):
    g, b, m = djb.ggx(ctx=ctx), djb.beckmann(ctx=ctx), djb.merl.from_table(tab, ctx=ctx)
    cases.append({"ggx.eval": lambda n, g=g: g.eval(i_all[:n], o_all[:n], p), "beckmann.sample": lambda n, b=b: b.sample(u1[:n], u2[:n], o_all[:n], p),
                  "merl.eval": lambda n, m=m: m.eval(i_all[:n], o_all[:n])})
print("# us per CALL of n units (DJB_MEM_HOST arrays, python ctypes overhead ~2 us included on both sides); host = one thread")
print("# %-6s" % "n" + "".join("%28s" % k for k in cases[0]))
print("# %-6s" % "" + "".join("%14s%14s" % ("gpu", "host") for _ in cases[0]))
for n in (16, 32, 64, 96, 128, 192, 256, 384, 512, 768, 1024, 2048, 4096):
    row = "  %-6d" % n
    for name in cases[0]:
        ts = []
        for side in (0, 1):
            f = cases[side][name]
            f(n); f(n)
            reps = 400 if n <= 512 else 100
            t0 = time.perf_counter()
            for _ in range(reps):
                f(n)
            ts.append((time.perf_counter() - t0) / reps * 1e6)
        row += "%14.1f%14.1f" % (ts[0], ts[1])
    print(row)
