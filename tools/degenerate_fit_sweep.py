#!/usr/bin/env python3
"""The tabular fitter on degenerate MERL tables (all zero, all below the horizon, constant, tiny, huge, a NaN texel, an Inf texel, one hot
texel) and odd resolutions against the oracle: tables, the Fresnel spline and both fits, value bits (NaN payloads aside).  The real
reference, built with its asserts on, stops in normalize_p22 (`nint > 0.0`, hdr:2296) on the zero / huge / NaN tables; the oracle -- and the
product -- compute what an NDEBUG build of it goes on to compute, and equal it bit for bit where it does not assert (constant, hot texel).
    python tools/degenerate_fit_sweep.py [--cpu]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oraclelib  # noqa: E402
from dj_brdf_amd import djb, synth  # noqa: E402

ctx = djb.cpu_context() if "--cpu" in sys.argv else djb.default_context(0)
O = oraclelib.oracle()


def vb(a):
    a = np.ascontiguousarray(np.asarray(a, np.float32))
    return np.where(np.isnan(a), np.uint32(0x7fc00000), a.view(np.uint32))


def raw_fit(tt):
    """the two fitted roughnesses as numbers (params::isotropic(alpha) asserts alpha > 0, in the reference and here)"""
    import ctypes as C
    a, g = C.c_float(), C.c_float()
    djb._lib.check(djb._lib.load().djb_tabular_fit(tt._h, C.byref(a), C.byref(g)))
    return a.value, g.value


base = synth.merl_table(0.3)
cases = {"all zero": np.zeros_like(base), "all negative (below the horizon)": -np.ones_like(base), "constant 0.2": np.full_like(base, 0.2 * 1500.0),
         "tiny 1e-30": np.full_like(base, 1e-30), "huge 1e30": np.full_like(base, 1e30)}
t = base.copy(); t.reshape(-1)[123456] = np.nan; cases["one NaN texel"] = t
t = base.copy(); t.reshape(-1)[654321] = np.inf; cases["one Inf texel"] = t
t = np.zeros_like(base); t.reshape(-1)[[1000, 1458000 + 1000, 2 * 1458000 + 1000]] = 5e4; cases["one hot texel"] = t
cases["ggx 0.3 + diffuse (control)"] = base
bad = 0
for name, tab in cases.items():
    for res, shadow in ((90, True), (17, False), (3, True)):
        try:
            tt = djb.tabular(djb.merl.from_table(tab, ctx=ctx), res, shadow, ctx=ctx)
        except djb.exc as e:
            print("%-34s res %-3d shadow %d   product raised: %s" % (name, res, shadow, e)); bad += 1; continue
        want = O.tabular_tables(O.tabular(O.merl_from_table(tab), res, shadow))
        got = {"p22": tt.get_p22v(), "sigma": tt.get_sigmav(), "cdf": tt.get_cdfv(), "qf": tt.get_qfv(), "fresnel": tt.get_fresnel().get_points(),
               "alpha_beckmann": [raw_fit(tt)[0]], "alpha_ggx": [raw_fit(tt)[1]]}
        diff = [k for k, v in got.items() if np.asarray(v).size != np.asarray(want[k]).size or not np.array_equal(vb(np.asarray(v).reshape(-1)), vb(np.asarray(want[k]).reshape(-1)))]
        print("%-34s res %-3d shadow %d   %s" % (name, res, shadow, "ok" if not diff else "MISMATCH in " + ", ".join(diff))); bad += bool(diff)
print("cases with a mismatch:", bad)
