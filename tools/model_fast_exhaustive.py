#!/usr/bin/env python3
"""GPU box: the decided fast tier of the sgd / abc chains (csrc/djb_fast_models.inc) against the exact chains over EVERY float polar
cosine of (0, 1] -- bit patterns 1 .. 0x3f800000, 1.07e9 floats -- for every published row (djb_selftest_model_fast, seed 0).
python tools/model_fast_exhaustive.py > profiles/r06/model_fast_exhaustive.txt"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dj_brdf_amd import djb, synth  # noqa: E402

ctx = djb.default_context(0)
N = 0x3f800000
t0 = time.time()
for kind in ("sgd", "abc"):
    tot = {"g1": 0, "g1_undecided": 0, "g1_mismatch": 0, "ndf": 0, "ndf_undecided": 0, "ndf_mismatch": 0}
    worst = (0.0, "")
    for name in synth.MERL_NAMES:
        r = djb.selftest_model_fast(getattr(djb, kind)(name, ctx=ctx), N, seed=0, first=1, ctx=ctx)
        for k in tot:
            tot[k] += r[k]
        und = (r["g1_undecided"] + r["ndf_undecided"]) / max(1, r["g1"] + r["ndf"])
        if und > worst[0]:
            worst = (und, name)
        if r["g1_mismatch"] or r["ndf_mismatch"]:
            print(f"{kind} {name}: MISMATCH {r}")
    print(f"{kind}: 100 rows x {N} floats: g1 {tot['g1']} values, {tot['g1_undecided']} undecided, {tot['g1_mismatch']} different; "
          f"ndf {tot['ndf']} values, {tot['ndf_undecided']} undecided, {tot['ndf_mismatch']} different; "
          f"largest undecided share of a row {worst[0]:.3g} ({worst[1]})", flush=True)
print(f"{time.time() - t0:.0f} s")
