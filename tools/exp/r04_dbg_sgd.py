import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
from dj_brdf_amd import djb, synth
import oraclelib
O = oraclelib.oracle()
ctx = djb.default_context(0)
rng = np.random.default_rng(6)
n = (1 << 16) + 1
i = synth.directions_aos(n, 31).copy(); o = synth.directions_aos(n, 32).copy()
i[:2000, 2] *= -1; o[2000:4000, 2] *= -1
i[4000:4100] = np.nan; o[4100:4200, 0] = np.inf
i[4200:6200] = o[4200:6200] * np.float32([-1, -1, 1]) + rng.normal(0, 1e-4, (2000, 3)).astype(np.float32)
o[6200:8200] = i[6200:8200]
i[8200:9200] *= np.float32(3.0)
i[9200:11200, 2] *= np.float32(1e-3); i[9200:11200] /= np.linalg.norm(i[9200:11200], axis=1, keepdims=True)
i = i.astype(np.float32); o = o.astype(np.float32)
soa = lambda a: torch.from_numpy(np.ascontiguousarray(a.T)).cuda()
di, do = soa(i), soa(o)
b, ob = djb.sgd("gold-metallic-paint", ctx=ctx), O.sgd("gold-metallic-paint")
want = O.eval(ob, i, o, None, "eval")
for on in (False, True):
    djb.set_contract_1e5(ctx, on)
    got = b.eval(di, do).cpu().numpy().T
    bad = np.flatnonzero((np.isnan(got) != np.isnan(want)).any(axis=1))
    print("contract", on, "NaN mismatches:", len(bad), bad[:10])
    for k in bad[:5]:
        print(k, i[k], o[k], got[k], want[k])
djb.set_contract_1e5(ctx, False)
