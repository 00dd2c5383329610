import torch
from dj_brdf_amd import djb, synth
ctx = djb.default_context(0)
mobj = djb.merl.from_table(synth.merl_table(0.3), ctx=ctx)
n = 1_000_000_000
i = djb.gen_directions(n, synth.SEED_I); o = djb.gen_directions(n, synth.SEED_O)
out = None
for rep in range(4):
    ctx.timer_start(); out = mobj.eval(i, o); ms = ctx.timer_stop_ms(); print("two-tier", ms, "ms", n/ms/1e6, "G/s", flush=True)
