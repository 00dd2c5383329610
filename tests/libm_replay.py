"""Helper of tests/test_host_libm.py, run in a subprocess (optionally under LD_PRELOAD of a perturbed libm): replays the
REAL reference's golden vectors (tests/golden/microfacet.npz, merl.npz) through the product's host path (or, with
`gpu`, through scalar-size host calls and a GPU batch of the same object) and prints one JSON line."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from dj_brdf_amd import djb, synth                      # noqa: E402
from golden_cases import MICROFACET_CASES              # noqa: E402
from test_gpu_parity import mk_fresnel, mk_params      # noqa: E402

G = os.path.join(ROOT, "tests", "golden")
mode = sys.argv[1] if len(sys.argv) > 1 else "cpu"
bits = lambda a: np.ascontiguousarray(a, np.float32).view(np.uint32)
out = {}
if mode == "cpu":
    ctx = djb.Context("cpu")
    out["status"] = djb.host_libm_status()
    g = np.load(os.path.join(G, "microfacet.npz"))
    i, o, u1, u2 = g["i"], g["o"], g["u1"], g["u2"]
    bad = 0
    for k, (ndf, fres, shadow, par) in enumerate(MICROFACET_CASES):
        b = getattr(djb, ndf)(mk_fresnel(fres), shadow, ctx=ctx)
        up = mk_params(par)
        for op in ("eval", "pdf"):
            bad += int(np.sum(bits(getattr(b, op)(i, o, up)) != bits(g[f"c{k}_{op}"])))
        bad += int(np.sum(bits(b.sample(u1, u2, o, up)) != bits(g[f"c{k}_sample"])))
    out["microfacet_values_differing_from_reference_goldens"] = bad
    m = np.load(os.path.join(G, "merl.npz"))
    out["merl_indices_differing"] = int(np.sum(djb.merl_index(m["i"], m["o"], ctx=ctx) != m["index"]))
else:
    # GPU box: the same object answers scalar-size (64-unit) host calls (host twin, on this thread) and one GPU batch
    ctx = djb.default_context(0)
    out["status"] = djb.host_libm_status()
    n = 1 << 12
    i, o = synth.directions_aos(n, synth.SEED_I), synth.directions_aos(n, synth.SEED_O)
    u1, u2 = synth.uniforms(n, synth.SEED_U1), synth.uniforms(n, synth.SEED_U2)
    diff = 0
    for b, p in ((djb.beckmann(djb.fresnel.schlick((1.0, 0.71, 0.29)), True, ctx=ctx), djb.microfacet.params.elliptic(0.2, 0.5, 0.7)),
                 (djb.merl.from_table(synth.merl_table_hashed(), ctx=ctx), None),
                 (djb.sgd("gold-metallic-paint", ctx=ctx), None)):
        big = b.eval(i, o, p)
        small = np.concatenate([b.eval(i[k:k + 64], o[k:k + 64], p) for k in range(0, n, 64)])
        diff += int(np.sum(bits(big) != bits(small)))
        if isinstance(b, djb.microfacet):
            bs = b.sample(u1, u2, o, p)
            ss = np.concatenate([b.sample(u1[k:k + 64], u2[k:k + 64], o[k:k + 64], p) for k in range(0, n, 64)])
            diff += int(np.sum(bits(bs) != bits(ss)))
    out["scalar_vs_batch_values_differing"] = diff
print(json.dumps(out))
