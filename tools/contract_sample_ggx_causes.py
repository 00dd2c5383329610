#!/usr/bin/env python3
"""Why samples leave the GGX contract sampler's fast path, by input family (needs a -DDJB_EXP_RARE_COUNT build: DJB_LIB_PATH=...)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from dj_brdf_amd import djb, synth
ctx = djb.default_context(0)
b = djb.ggx(ctx=ctx); P = djb.microfacet.params
n = 1 << 24
o = djb.gen_directions(n, synth.SEED_O, ctx=ctx)
u1 = djb.gen_uniforms(n, synth.SEED_U1, ctx=ctx); u2 = djb.gen_uniforms(n, synth.SEED_U2, ctx=ctx)
djb.set_contract_1e5(ctx, True)
for name, p in (("isotropic(1.0)", P.isotropic(1.0)), ("elliptic(0.2,0.5,0.7)", P.elliptic(0.2, 0.5, 0.7))):
    for fam in ("bench", "u1 tail lo", "u1 tail hi", "u2 tail lo", "u2 tail hi"):
        a1, a2 = u1, u2
        if fam == "u1 tail lo": a1 = 1e-4 * u1
        if fam == "u1 tail hi": a1 = 1 - 1e-4 * (1 - u1)
        if fam == "u2 tail lo": a2 = 1e-3 * u2
        if fam == "u2 tail hi": a2 = 1 - 1e-3 * (1 - u2)
        print("==", name, fam, flush=True); sys.stderr.flush()
        keep = b.sample(a1.contiguous(), a2.contiguous(), o, p); del keep
        torch.cuda.synchronize()
