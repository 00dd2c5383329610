#!/usr/bin/env python3
"""Calibration of the two-tier MERL kernel's guard bands (run on the GPU box): max |fp32 estimate -
reference| / band per coordinate and the certain/ambiguous/mismatch counters over N x 2.5e8 pairs
(DESIGN.md 4.2).  PYTHONPATH=. python tools/calibrate_merl_guard.py [N]"""
import sys, time, numpy as np, torch
from dj_brdf_amd import djb, synth
ctx = djb.default_context(0)
n = 250_000_000
tot = {"special":0,"ambiguous":0,"mismatch":0,"certain":0}; mx = np.zeros(3)
for c in range(int(sys.argv[1]) if len(sys.argv)>1 else 4):
    i = djb.gen_directions(n, synth.SEED_I + 17*c, start=c*n); o = djb.gen_directions(n, synth.SEED_O + 31*c, start=c*n)
    s = djb.merl_guard_stats(i, o, ctx=ctx)
    mx = np.maximum(mx, s["max_ratio"])
    for k in tot: tot[k] += s[k]
    print(c, s, flush=True)
    del i, o
N = sum(tot[k] for k in ("special","ambiguous","certain"))
print("TOTAL", N, "max_ratio", mx, {k: v / N for k, v in tot.items()}, "mismatch", tot["mismatch"])
# adversarial families: near-specular (o ~ reflect(i)), near-backscatter (o ~ i), grazing
m = 50_000_000
base = djb.gen_directions(m, 99)
for name, f in [("backscatter", lambda b: b + 1e-3 * djb.gen_directions(m, 5)),
                ("backscatter_tiny", lambda b: b + 1e-5 * djb.gen_directions(m, 6)),
                ("mirror", lambda b: torch.stack([-b[0], -b[1], b[2]]) + 1e-3 * djb.gen_directions(m, 7)),
                ("grazing", lambda b: torch.stack([b[0], b[1], 1e-3 * b[2]])),
                ("identical", lambda b: b.clone())]:
    o = f(base); o = o / o.norm(dim=0, keepdim=True)
    s = djb.merl_guard_stats(base, o.contiguous(), ctx=ctx)
    print(name, s, flush=True)
# timing + full check of the two-tier kernel against the exact kernel
tab = synth.merl_table(0.3)
mobj = djb.merl.from_table(tab, ctx=ctx)
n = 500_000_000
i = djb.gen_directions(n, synth.SEED_I); o = djb.gen_directions(n, synth.SEED_O)
for rep in range(2):
    ctx.timer_start(); a = mobj.eval(i, o); ms = ctx.timer_stop_ms(); print("two-tier", ms, "ms", n/ms/1e6, "G/s")
djb.set_merl_exact_only(ctx, True)
ctx.timer_start(); b = mobj.eval(i, o); ms = ctx.timer_stop_ms(); print("exact", ms, "ms", n/ms/1e6, "G/s")
djb.set_merl_exact_only(ctx, False)
print("two-tier == exact:", bool(torch.equal(a, b)), int((a != b).any(dim=0).sum()))
