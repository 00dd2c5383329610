#!/usr/bin/env python3
"""djb_selftest_contract_sample on the GGX lobe over lobes x families (profiles/r04/contract_sample_ggx.txt)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from dj_brdf_amd import djb
from test_gpu_contract import SAMPLE_PARAMS, mk_params
ctx = djb.default_context(0)
for ndf in ("ggx",):
    b = getattr(djb, ndf)(ctx=ctx)
    for p in [None, ("elliptic", 0.02, 0.02, 0.0), ("elliptic", 0.2, 0.5, 0.7), ("pdfparams", 0.4, 0.25, 0.6, 0.1, -0.2), ("elliptic", 0.05, 0.8, 0.3), ("elliptic", 1.0, 1.0, 0.0)]:
        for family in range(5):
            r = djb.selftest_contract_sample(b, mk_params(p), n=1 << 26, seed=21 + family, family=family, ctx=ctx)
            print(ndf, p, family, "max %.2e used %.3f exact %.4f out %d" % (r["max_abs_dir"], r["bound_used"], r["exact_path"] / r["samples"], r["outside_1e5"]))
