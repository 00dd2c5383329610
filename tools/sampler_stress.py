#!/usr/bin/env python3
"""Randomised stress of the two-path Beckmann sampling kernel (djb_kernels_sample.hip) against the oracle, on the GPU box:
random lobes far outside the bench's (alpha 1e-4 .. 10, correlation to +-0.99, tilted means), the adversarial inputs of
tests/test_gpu_parity.py::_beckmann_sampler_cases (every reason a sample leaves the common path) plus un-normalised and
extreme-magnitude directions, device-resident dense batches (the kernel's fast path) and host batches.  Every bit of
sample() and evalp_is() must match.   PYTHONPATH=. python tools/sampler_stress.py [lobes] [n_bulk] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import oraclelib
from dj_brdf_amd import djb
from test_gpu_parity import _beckmann_sampler_cases, _same_bits

lobes = int(sys.argv[1]) if len(sys.argv) > 1 else 40
n_bulk = int(float(sys.argv[2])) if len(sys.argv) > 2 else 400_001
rng = np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 31)
O = oraclelib.oracle(); ctx = djb.default_context(0); dev = torch.device("cuda", 0)
o, u1, u2 = _beckmann_sampler_cases(n_bulk)
# un-normalised directions over the whole float range (the guarded 1/sqrt's domain is (1e-30, 1e30)) and a few non-finite ones
k = rng.permutation(n_bulk)[:30000]
o[k[:24000]] *= (10.0 ** rng.uniform(-22, 22, 24000)).astype(np.float32)[:, None]
o[k[24000:27000]] *= np.float32(1e-38)
o[k[27000:29000], 2] = np.float32(np.inf)
o[k[29000:], 0] = -np.float32(np.inf)
tu1, tu2 = torch.as_tensor(u1, device=dev), torch.as_tensor(u2, device=dev)
od = torch.as_tensor(np.ascontiguousarray(o.T), device=dev)
g = djb.beckmann(djb.fresnel.schlick((1.0, 0.71, 0.29)), True, ctx=ctx)
og = O.microfacet("beckmann", ("schlick", 1.0, 0.71, 0.29), True)
total = differ = 0
for r in range(lobes):
    if r % 2 == 0:
        a1, a2 = (float(np.float32(10.0 ** rng.uniform(-4, 1))) for _ in range(2))
        rho = float(np.float32(rng.uniform(-0.99, 0.99)))
        pp, up = ("elliptic", a1, a2, rho), djb.microfacet.params.elliptic(a1, a2, rho)
    else:
        v = [float(np.float32(x)) for x in (10.0 ** rng.uniform(-3, 0.7), 10.0 ** rng.uniform(-3, 0.7), rng.uniform(-0.95, 0.95), rng.uniform(-2, 2), rng.uniform(-2, 2))]
        pp, up = ("pdfparams", *v), djb.microfacet.params.pdfparams(*v)
    want = O.sample(og, u1, u2, o, pp)
    ww, wi, wpdf = O.evalp_is(og, u1, u2, o, pp)
    got = g.sample(tu1, tu2, od, up).cpu().numpy().T
    w, gi, pdf = g.evalp_is(tu1, tu2, od, up)
    goth = g.sample(u1, u2, o, up)
    checks = [("sample dense", got, want), ("sample host", goth, want), ("is dir", gi.cpu().numpy().T, wi), ("is weight", w.cpu().numpy().T, ww), ("is pdf", pdf.cpu().numpy(), wpdf)]
    line = []
    for tag, a, b in checks:
        same = _same_bits(a, b)
        total += same.size; d = int((~same).sum()); differ += d
        if d:
            idx = np.flatnonzero(~(same.all(axis=1) if same.ndim == 2 else same))[:3]
            line.append(f"{tag}: {d} differ, first {idx} o={o[idx]} u1={u1[idx]} u2={u2[idx]}")
    print(f"lobe {r:3d} {pp}: " + ("identical" if not line else "; ".join(line)), flush=True)
print(f"values compared {total:.4g}, not bit-identical {differ}")
sys.exit(1 if differ else 0)
