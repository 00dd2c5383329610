#!/usr/bin/env python3
"""Long run of djb_selftest_guarded_math (on the GPU box): the guarded fp64 shortcuts against the exact
double sequences on N x 4e9 hash-generated inputs.  Every *_mismatch total must be 0.
PYTHONPATH=. python tools/selftest_guarded.py [N]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dj_brdf_amd import djb
ctx = djb.default_context(0)
tot, N = {}, 0
for s in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    n = 4_000_000_000
    t0 = time.time(); r = djb.selftest_guarded_math(n, seed=1000 + s, ctx=ctx); dt = time.time() - t0
    for k, v in r.items():
        tot[k] = tot.get(k, 0) + v
    N += n
    print(s, r, "%.2fs" % dt, flush=True)
print("TOTAL n=%d " % N + " ".join(f"{k}={v}" + (f" ({v / N:.2e})" if k.endswith("fallback") else "") for k, v in tot.items()))
