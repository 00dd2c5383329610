#!/usr/bin/env python3
"""How far apart are the reference's two numerical variants?  (TEST INFRASTRUCTURE: loads oracle/_ref, build container only.)

The reference header's unqualified acos / atan2 / cos / sin / sqrt / exp inside namespace djb resolve to the DOUBLE C functions when the
translation unit includes only <cmath> (examples/merl_params.cpp:10-16 -- the variant the kernels, the oracle and the goldens reproduce,
oracle/_ref/libdjb_ref.so) and to the FLOAT overloads when <math.h> comes first (tests/nrm_utia.cpp:8; a Mitsuba build of the plugins --
oracle/_ref/libdjb_ref_mathh.so, same shim built with `-include math.h`).  This script runs both builds on the golden-test inputs and
prints, per operator, the share of outputs that differ at all, the share beyond 1e-5 relative, the largest relative difference -- and for
MERL the share of look-ups that land in another BIN.  `python tests/ref_mathh_distance.py [n]` -> JSON on stdout
(profiles/r06/ref_mathh_distance.json is that output; INTEGRATION.md section 5 quotes it; tests/test_ref_mathh_variant.py re-derives it)."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
import oraclelib
from dj_brdf_amd import synth


def compare(a, b):
    a = np.asarray(a, np.float64).reshape(-1); b = np.asarray(b, np.float64).reshape(-1)
    both_nan = np.isnan(a) & np.isnan(b)
    differ = ~both_nan & ~(a == b)
    with np.errstate(divide="ignore", invalid="ignore"):
        rel = np.where(differ, np.abs(a - b) / np.maximum(np.maximum(np.abs(a), np.abs(b)), 1e-30), 0.0)
    rel = np.where(np.isfinite(rel), rel, np.where(differ, np.inf, 0.0))
    return {"values": int(a.size), "differ": float(differ.mean()), "beyond_1e-5": float((rel > 1e-5).mean()),
            "max_rel": float(rel.max()) if rel.size else 0.0}


def measure(n=200_000):
    ref_dir = os.path.join(ROOT, "oracle", "_ref")
    A = oraclelib.CheckerLib(os.path.join(ref_dir, "libdjb_ref.so"), "ref_")
    B = oraclelib.CheckerLib(os.path.join(ref_dir, "libdjb_ref_mathh.so"), "ref_")
    i, o = synth.directions_aos(n, synth.SEED_I), synth.directions_aos(n, synth.SEED_O)
    u1, u2 = synth.uniforms(n, synth.SEED_U1), synth.uniforms(n, synth.SEED_U2)
    out = {"pairs": n, "inputs": "synth.directions_aos / synth.uniforms (the golden-test generators), seeds SEED_I / SEED_O / SEED_U1 / SEED_U2"}
    ia, ib = A.merl_index(i, o), B.merl_index(i, o)
    out["merl_index"] = {"values": n, "other_bin": float((ia != ib).mean()), "count": int((ia != ib).sum())}
    path = "/tmp/djb_mathh_merl.binary"
    synth.write_merl_binary(path, synth.merl_table(0.3))
    out["merl.eval"] = compare(A.eval(A.merl(path), i, o), B.eval(B.merl(path), i, o))
    upath = "/tmp/djb_mathh_utia.bin"
    np.random.default_rng(11).uniform(0.0, 120.0, size=3 * 288 * 288).tofile(upath)
    out["utia.eval"] = compare(A.eval(A.utia(upath), i, o), B.eval(B.utia(upath), i, o))
    par = ("elliptic", 0.2, 0.5, 0.7)
    for ndf in ("ggx", "beckmann"):
        fa, fb = A.microfacet(ndf, ("schlick", 1.0, 0.71, 0.29), True), B.microfacet(ndf, ("schlick", 1.0, 0.71, 0.29), True)
        for op in ("eval", "pdf"):
            out[f"{ndf}.{op}"] = compare(A.eval(fa, i, o, par, op), B.eval(fb, i, o, par, op))
        out[f"{ndf}.sample"] = compare(A.sample(fa, u1, u2, o, par), B.sample(fb, u1, u2, o, par))
    for model in ("sgd", "abc"):
        out[f"{model}.eval"] = compare(A.eval(getattr(A, model)("gold-metallic-paint"), i, o), B.eval(getattr(B, model)("gold-metallic-paint"), i, o))
    ta, tb = A.tabular(A.merl(path), 90, True), B.tabular(B.merl(path), 90, True)
    TA, TB = A.tabular_tables(ta), B.tabular_tables(tb)
    for k in ("p22", "sigma", "cdf"):
        out[f"tabular(merl, 90).{k}"] = compare(TA[k], TB[k])
    out["tabular(merl, 90).alphas"] = {"beckmann": [float(TA["alpha_beckmann"]), float(TB["alpha_beckmann"])], "ggx": [float(TA["alpha_ggx"]), float(TB["alpha_ggx"])]}
    out["tabular(merl, 90).eval"] = compare(A.eval(ta, i, o), B.eval(tb, i, o))
    return out


if __name__ == "__main__":
    print(json.dumps(measure(int(sys.argv[1]) if len(sys.argv) > 1 else 200_000), indent=1))
