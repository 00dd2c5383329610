#!/usr/bin/env python3
"""Launch-bound inner loops: K small device-memory batches issued eagerly (one ctypes call + one or two kernel launches each) against
the same K calls captured once into a hipGraph (torch.cuda.CUDAGraph on the context's stream) and replayed.
    python tools/graph_replay_rate.py > gpurun_out/graph_replay.txt          (on the GPU box)"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from dj_brdf_amd import djb, synth  # noqa: E402

ctx = djb.default_context(0)
lib = djb._lib.load()
K = 256
print(f"{K} calls per round, median of 20 rounds; us per call")
print("%-22s %9s %12s %12s %8s" % ("kind", "n", "eager", "graph", "ratio"))
for kind in ("ggx eval+pdf", "merl eval", "beckmann sample"):
    for n in (1 << 10, 1 << 12, 1 << 14, 1 << 16, 1 << 18):
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            i = djb.gen_directions(n, synth.SEED_I, ctx=ctx); o = djb.gen_directions(n, synth.SEED_O, ctx=ctx)
            vi, vo = djb._Vec(i), djb._Vec(o)
            out = torch.zeros((3, n), dtype=torch.float32, device=i.device); vout = djb._Vec(out)
            pdf = torch.zeros(n, dtype=torch.float32, device=i.device)
            if kind == "ggx eval+pdf":
                obj = djb.ggx(djb.fresnel.schlick((1.0, 0.71, 0.29)), True, ctx=ctx); p = djb.microfacet.params.isotropic(0.3)
                def call():
                    lib.djb_eval_pdf_batch(ctx._h, obj._h, C.c_int64(n), C.byref(vi.view), C.byref(vo.view), djb._params_ptr(p), C.c_int(0),
                                           C.byref(vout.view), C.c_void_p(pdf.data_ptr()), C.c_int(0))
            elif kind == "merl eval":
                obj = djb.merl.from_table(synth.merl_table(0.3), ctx=ctx)
                def call():
                    lib.djb_eval_batch(ctx._h, obj._h, C.c_int64(n), C.byref(vi.view), C.byref(vo.view), None, C.byref(vout.view), C.c_int(0))
            else:
                obj = djb.beckmann(ctx=ctx); p = djb.microfacet.params.elliptic(0.2, 0.5, 0.7)
                def call():
                    lib.djb_sample_rng_batch(ctx._h, obj._h, C.c_int64(n), C.c_uint32(synth.SEED_U1), C.c_uint32(synth.SEED_U2), C.c_uint64(0),
                                             C.byref(vo.view), djb._params_ptr(p), C.byref(vout.view))
            for _ in range(3):
                call()
            side.synchronize()
            te = []
            for _ in range(20):
                t0 = time.perf_counter()
                for _ in range(K):
                    call()
                side.synchronize()
                te.append(time.perf_counter() - t0)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for _ in range(K):
                call()
        g.replay(); torch.cuda.synchronize()
        tg = []
        for _ in range(20):
            t0 = time.perf_counter()
            g.replay()
            torch.cuda.synchronize()
            tg.append(time.perf_counter() - t0)
        e, r = sorted(te)[10] / K * 1e6, sorted(tg)[10] / K * 1e6
        print("%-22s %9d %12.2f %12.2f %8.2f" % (kind, n, e, r, e / r))
