import time, numpy as np, torch
from dj_brdf_amd import djb, synth
ctx = djb.default_context(0)
def timeit(f, reps=5):
    f(); torch.cuda.synchronize()
    ts=[]
    for _ in range(reps):
        ctx.timer_start(); f(); ts.append(ctx.timer_stop_ms())
    return np.median(ts)
for n in (100_000_000,):
    i = djb.gen_directions(n, synth.SEED_I); o = djb.gen_directions(n, synth.SEED_O)
    g = djb.ggx(djb.fresnel.ideal(), True); p = djb.microfacet.params.isotropic(0.3)
    for name, f, byt in [("ggx eval", lambda: g.eval(i,o,p), 36), ("ggx pdf", lambda: g.pdf(i,o,p), 28), ("ggx eval+pdf", lambda: g.eval_pdf(i,o,p), 40)]:
        ms = timeit(f); print(f"{name}: n={n} {ms:.3f} ms  {n/ms/1e6:.2f} G/s  {n*byt/ms/1e6:.1f} GB/s")
    gs = djb.ggx(djb.fresnel.schlick((1.0,0.71,0.29)), True)
    ms = timeit(lambda: gs.eval(i,o,p)); print(f"ggx schlick eval: {ms:.3f} ms {n/ms/1e6:.2f} G/s")
    bk = djb.beckmann(djb.fresnel.ideal(), True); pe = djb.microfacet.params.elliptic(0.2,0.5,0.7)
    ms = timeit(lambda: bk.eval(i,o,pe)); print(f"beckmann eval: {ms:.3f} ms {n/ms/1e6:.2f} G/s")
    ms = timeit(lambda: bk.sample_rng(synth.SEED_U1, synth.SEED_U2, o, pe)); print(f"beckmann sample rng: {ms:.3f} ms {n/ms/1e6:.2f} G/s  {n*24/ms/1e6:.1f} GB/s")
    ms = timeit(lambda: g.sample_rng(synth.SEED_U1, synth.SEED_U2, o, p)); print(f"ggx sample rng: {ms:.3f} ms {n/ms/1e6:.2f} G/s")
    m = djb.merl.from_table(synth.merl_table(0.3))
    ms = timeit(lambda: m.eval(i,o)); print(f"merl eval: {ms:.3f} ms {n/ms/1e6:.2f} G/s {n*36/ms/1e6:.1f} GB/s")
    ms = timeit(lambda: djb.gen_directions(n, 1)); print(f"gen dirs: {ms:.3f} ms {n*12/ms/1e6:.1f} GB/s")
tabs = [synth.merl_table(*synth.material_recipe(k)) for k in range(4)]
t0=time.time(); ab, ag = djb.fit_merl_batch(tabs); t1=time.time(); print("fit 4 mats (incl upload)", t1-t0, ab, ag)
src = djb.merl.from_table(tabs[0])
for _ in range(3):
    t0=time.time(); t = djb.tabular(src, 90); t1=time.time(); print("tabular(merl,90) wall", t1-t0)
