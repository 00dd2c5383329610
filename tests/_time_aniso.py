import time, numpy as np, torch
from dj_brdf_amd import djb, synth
ctx = djb.default_context(0)
rng = np.random.default_rng(11)
u = djb.utia.from_table(rng.uniform(0.0, 120.0, size=3*288*288), ctx=ctx)
m = djb.merl.from_table(synth.merl_table(0.3), ctx=ctx)
for name, src in (("utia", u), ("merl", m), ("ggx", djb.ggx(ctx=ctx))):
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        t = djb.tabular_anisotropic(src, 90, 90, True, ctx=ctx)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    fb = djb.tabular_anisotropic.fit_beckmann_parameters(t).get_pdfparams()
    print(f"tabular_anisotropic({name}, 90, 90): {dt*1e3:.2f} ms  (N = 8010, kernel matrix never stored)  beckmann fit {np.round(fb, 4)}")
