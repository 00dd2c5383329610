#!/usr/bin/env python3
"""Bit-exactness report (run on the GPU box): fraction of eval / pdf outputs of the HIP path that are
bit-identical to the CPU oracle, per NDF x Fresnel x params, on 2^20 pairs.  PYTHONPATH=. python tests/exactness_report.py"""
import numpy as np, sys
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import oraclelib
from dj_brdf_amd import djb, synth
from test_gpu_parity import FRESNELS, PARAMS, mk_fresnel, mk_params
O = oraclelib.oracle(); ctx = djb.default_context(0)
n = 1 << 20
i = synth.directions_aos(n, synth.SEED_I); o = synth.directions_aos(n, synth.SEED_O)
for ndf in ("ggx", "beckmann"):
    for fres in FRESNELS:
        g = getattr(djb, ndf)(mk_fresnel(fres), True, ctx=ctx); ob = O.microfacet(ndf, fres, True)
        for p in PARAMS:
            for op in ("eval", "pdf"):
                a = getattr(g, op)(i, o, mk_params(p)); b = O.eval(ob, i, o, p, op)
                ex = np.mean(a.view(np.uint32) == b.view(np.uint32))
                rel = np.max(np.abs(a.astype(np.float64) - b) / np.maximum(np.abs(b), 1e-25))
                print(f"{ndf:9s} {fres[0]:12s} {str(p)[:28]:28s} {op:5s} bit-exact {ex:.7f} max rel {rel:.2e}")
