#!/usr/bin/env python3
"""DJB_OPT_CONTRACT_1E5 for Beckmann `sample` (run on the GPU box): (1) djb_selftest_contract_sample over lobes x input
families -- largest component difference against the bit-exact per-sample code, share of the samples handed to the exact
path, samples outside 1e-5 (must be 0); (2) sample_rng over 1e9 directions with the option off / on.
    PYTHONPATH=. python tools/contract_sample_probe.py [n_selftest] > profiles/r04/contract_sample.txt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dj_brdf_amd import djb, synth
ctx = djb.default_context(0)
n_st = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1 << 28
b = djb.beckmann(ctx=ctx)
P = djb.microfacet.params
lobes = [("elliptic(0.2,0.5,0.7)", P.elliptic(0.2, 0.5, 0.7)), ("isotropic(0.3)", P.isotropic(0.3)), ("isotropic(0.02)", P.isotropic(0.02)),
         ("isotropic(1.0)", P.isotropic(1.0)), ("elliptic(0.05,0.8,0.3)", P.elliptic(0.05, 0.8, 0.3)), ("pdfparams(0.4,0.25,0.6,0.1,-0.2)", P.pdfparams(0.4, 0.25, 0.6, 0.1, -0.2)),
         ("pdfparams(3,5,-0.9,0,0)", P.pdfparams(3.0, 5.0, -0.9, 0.0, 0.0))]
fam = ["bench", "grazing", "near-normal", "uniform tails", "un-normalised"]
print(f"# djb_selftest_contract_sample, {n_st:.3g} samples per cell: max |component diff| among kept samples / share handed to the exact path / kept samples outside 1e-5 / largest difference : bound")
worst, bad, used = 0.0, 0, 0.0
for name, p in lobes:
    row = []
    for f in range(5):
        r = djb.selftest_contract_sample(b, p, n=n_st, seed=11 + f, family=f, ctx=ctx)
        worst = max(worst, r["max_abs_dir"]); bad += r["outside_1e5"]
        used = max(used, r["bound_used"])
        row.append(f"{fam[f]}: {r['max_abs_dir']:.2e} / {100.0 * r['exact_path'] / r['samples']:.2f} % / {r['outside_1e5']} / {r['bound_used']:.2f}")
    print(f"{name:34s} " + "   ".join(row))
print(f"# worst component difference {worst:.3e}, samples outside 1e-5: {bad}, largest share of the per-sample bound used: {used:.2f}")
n = 1_000_000_000
o = djb.gen_directions(n, synth.SEED_O, ctx=ctx)
p = P.elliptic(0.2, 0.5, 0.7)
for on in (False, True, False, True):
    djb.set_contract_1e5(ctx, on)
    for _ in range(3): keep = b.sample_rng(synth.SEED_U1, synth.SEED_U2, o, p); del keep
    torch.cuda.synchronize(); ctx.timer_start()
    for _ in range(5): keep = b.sample_rng(synth.SEED_U1, synth.SEED_U2, o, p); del keep
    ms = ctx.timer_stop_ms() / 5
    print(f"sample_rng 1e9, elliptic(0.2,0.5,0.7), contract {'on ' if on else 'off'}: {ms:7.3f} ms -> {n / ms / 1e6:6.1f} G samples/s = {24 * n / ms / 1e6 / 8000:.3f} of 8 TB/s")
djb.set_contract_1e5(ctx, False)
