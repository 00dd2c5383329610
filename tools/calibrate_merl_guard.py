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


def unit(v):
    return (v / v.norm(dim=0, keepdim=True)).contiguous()


def on_cone(axis, theta_deg, phi):
    """unit vectors at angle theta from `axis` (any frame completion), azimuth phi"""
    t = torch.deg2rad(theta_deg)
    ref = torch.zeros_like(axis); ref[0] = 1.0
    ref = torch.where((axis[0].abs() > 0.9).unsqueeze(0), torch.stack([torch.zeros_like(axis[0]), torch.ones_like(axis[0]), torch.zeros_like(axis[0])]), ref)
    e1 = unit(torch.linalg.cross(axis, ref, dim=0)); e2 = torch.linalg.cross(axis, e1, dim=0)
    return unit(torch.cos(t) * axis + torch.sin(t) * (torch.cos(phi) * e1 + torch.sin(phi) * e2))


g = torch.Generator(device=base.device); g.manual_seed(1234)
R = lambda *shape: torch.rand(*shape, device=base.device, generator=g)
more = {}
# both directions within a few degrees of the normal (theta_h ~ 0: the sqrt-spaced bins are narrowest there)
more["near_normal"] = (unit(torch.stack([1e-2 * (R(m) - .5), 1e-2 * (R(m) - .5), torch.ones(m, device=base.device)])),
                       unit(torch.stack([3e-2 * (R(m) - .5), 3e-2 * (R(m) - .5), torch.ones(m, device=base.device)])))
# uniform on the whole sphere: half of the directions are below the horizon
sph = lambda: unit(torch.randn(3, m, device=base.device, generator=g))
more["full_sphere"] = (sph(), sph())
# un-normalised inputs (the reference never normalises i / o)
more["lengths_0.5_to_2"] = (base * (0.5 + 1.5 * R(m)), djb.gen_directions(m, 11) * (0.5 + 1.5 * R(m)))
# theta_d within 1e-5 degrees of an integer number of degrees: o = i rotated about h by construction
h = djb.gen_directions(m, 21)
td = torch.randint(1, 89, (m,), device=base.device).float() + (R(m) - .5) * 2e-5
ph = 2 * torch.pi * R(m)
ii = on_cone(h, td, ph)
oo = unit(2 * (ii * h).sum(0, keepdim=True) * h - ii)
more["theta_d_on_bin_edges"] = (ii, oo)
# theta_h on the sqrt-spaced bin edges: theta_h = k^2/90 degrees
k = torch.randint(1, 89, (m,), device=base.device).float()
th = k * k / 90.0 + (R(m) - .5) * 2e-5
hh = on_cone(torch.stack([torch.zeros(m, device=base.device), torch.zeros(m, device=base.device), torch.ones(m, device=base.device)]), th, 2 * torch.pi * R(m))
ii = on_cone(hh, 5 + 70 * R(m), 2 * torch.pi * R(m))
oo = unit(2 * (ii * hh).sum(0, keepdim=True) * hh - ii)
more["theta_h_on_bin_edges"] = (ii, oo)
for name, (a, b) in more.items():
    s = djb.merl_guard_stats(a.contiguous(), b.contiguous(), ctx=ctx)
    print(name, s, flush=True)
    assert s["mismatch"] == 0, name
if len(sys.argv) > 2 and sys.argv[2] == "families":
    sys.exit(0)
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
