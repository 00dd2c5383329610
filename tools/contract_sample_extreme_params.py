#!/usr/bin/env python3
"""The contract-mode samplers (djb_selftest_contract_sample) at the corners of their parameter domain (1e-3 <= ax, ay <= 100, |rho| <= 0.99,
|tx|, |ty| <= 10): nothing kept may be outside 1e-5, the bound must bound.  python tools/contract_sample_extreme_params.py  (GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from dj_brdf_amd import djb
from test_gpu_contract import mk_params
ctx = djb.default_context(0)
for ndf in ("ggx", "beckmann"):
    b = getattr(djb, ndf)(ctx=ctx)
    for p in [("pdfparams", 0.3, 0.3, 0.99, 0.0, 0.0), ("pdfparams", 0.3, 0.3, -0.99, 5.0, -3.0), ("pdfparams", 50.0, 0.002, 0.5, 0.0, 0.0), ("pdfparams", 0.002, 80.0, -0.9, 9.0, 9.0),
              ("elliptic", 0.001, 0.001, 0.0), ("elliptic", 100.0, 100.0, 0.0), ("elliptic", 0.0011, 99.0, 1.0)]:
        for family in range(5):
            try:
                r = djb.selftest_contract_sample(b, mk_params(p), n=1 << 24, seed=31 + family, family=family, ctx=ctx)
            except djb.exc:
                print(ndf, p, "outside the sampler's domain"); break
            flag = "" if (r["outside_1e5"] == 0 and r["bound_used"] < 1.0) else "   <-- ATTENTION"
            print(ndf, p, family, "max %.2e used %.3f exact %.4f out %d%s" % (r["max_abs_dir"], r["bound_used"], r["exact_path"] / r["samples"], r["outside_1e5"], flag))
