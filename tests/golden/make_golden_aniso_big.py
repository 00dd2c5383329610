#!/usr/bin/env python3
"""tests/golden/aniso_big.npz from the REAL reference (build container only; needs /root/reference):
djb::tabular_anisotropic at 90 x 90 (the reference's own size: an 8010^2-double kernel matrix, 513 MB, ~8 s per
fit), UTIA-sourced fits, and two fits whose conditional quantile table comes up short (dj_brdf.h:3005-3034).

    python tests/golden/make_golden_aniso_big.py

Stored per case: p22 / sigma grids, Fresnel points, both 5-parameter fits, the six sampling queries, eval / pdf of
the fitted object, sample().  For the short-row cases the qf2 queries are restricted to taps the reference's
m_qf2 really holds (what lies past its end is undefined behaviour there) -- see `defined_qf2_queries`."""
import os
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oraclelib  # noqa: E402
from dj_brdf_amd import synth  # noqa: E402
from golden_cases import ANISO_BIG_CASES, N_ANISO_BIG  # noqa: E402


def big_source(L, src, tmpdir):
    """source BRDF of an ANISO_BIG_CASES entry on checker library L (reference: through files)."""
    if src[0] == "utia":
        path = os.path.join(tmpdir, "smooth_utia.bin")
        synth.utia_table_smooth().tofile(path)
        return L.utia(path)
    tab = synth.merl_table(*src[1:]) if src[0] == "merl" else synth.merl_table_grazing(src[1])
    if L.prefix == "o_":
        return L.merl_from_table(tab)
    path = os.path.join(tmpdir, f"src_{src[0]}.binary")
    synth.write_merl_binary(path, tab)
    return L.merl(path)


def defined_qf2_queries(elev, azim, entries, n, seed=0xA51):
    """(u, phi) whose four eval2d taps lie at indices < entries of an elev-strided read (dj_brdf.h:1220-1249):
    rows j1, j2 = floor(phi-coordinate), +1 (wrapped) must satisfy elev*j2 + elev - 1 < entries."""
    rows_ok = entries // elev            # rows 0 .. rows_ok-1 are complete
    assert rows_ok >= 2
    u = synth.uniforms(n, seed)
    r = synth.uniforms(n, seed + 1) * np.float32(rows_ok - 1)          # phi coordinate in [0, rows_ok - 1): j2 <= rows_ok - 1
    # u2 = phi / (2 pi); t2 = u2 * azim - u2  -> choose phi so that t2 = r
    phi = (r.astype(np.float64) / (azim - 1) * 2.0 * np.pi).astype(np.float32)
    t2 = (phi.astype(np.float64) / (2 * np.pi)).astype(np.float32)
    t2 = t2 * np.float32(azim) - t2
    keep = np.floor(t2).astype(int) + 1 <= rows_ok - 1
    return u[keep], phi[keep]


def main():
    R = oraclelib.reference()
    assert R is not None, "needs /root/reference (build container)"
    O = oraclelib.oracle()
    n = N_ANISO_BIG
    i = synth.directions_aos(n, synth.SEED_I, start=70000)
    o = synth.directions_aos(n, synth.SEED_O, start=70000)
    u1 = synth.uniforms(n, synth.SEED_U1, start=70000)
    u2 = synth.uniforms(n, synth.SEED_U2, start=70000)
    out = {"i": i, "o": o, "u1": u1, "u2": u2}
    tmp = tempfile.mkdtemp(prefix="djb_golden_aniso_big_")
    try:
        for name, (src, elev, azim, shadow) in ANISO_BIG_CASES.items():
            t = R.tabular_anisotropic(big_source(R, src, tmp), elev, azim, shadow)
            for k, v in R.aniso_tables(t).items():
                out[f"{name}_{k}"] = v
            phi, th = (u1 * np.float32(6.2)).astype(np.float32), (u2 * np.float32(1.5)).astype(np.float32)
            short = name.startswith("a_short")
            for q, args in (("pdf1", (phi,)), ("cdf1", (phi,)), ("qf1", (u1,)), ("pdf2", (th, phi)), ("cdf2", (th, phi))):
                out[f"{name}_{q}"] = R.aniso_query(t, q, *args)
            if short:
                # how many entries the reference's vector holds: from the (reference-pinned) restatement
                ot = O.tabular_anisotropic(big_source(O, src, tmp), elev, azim, shadow)
                entries = O.aniso_sampling_tables(ot)["qf2_entries"]
                assert entries < elev * azim, f"{name}: no short row"
                qu, qphi = defined_qf2_queries(elev, azim, entries, 4 * n)
                out[f"{name}_qf2_entries"] = np.array([entries], np.int32)
                out[f"{name}_qf2_u"], out[f"{name}_qf2_phi"] = qu, qphi
                out[f"{name}_qf2"] = R.aniso_query(t, "qf2", qu, qphi)
            else:
                out[f"{name}_qf2"] = R.aniso_query(t, "qf2", u2, phi)
                out[f"{name}_sample"] = R.sample(t, u1, u2, o)
            for op in ("eval", "pdf"):
                out[f"{name}_{op}"] = R.eval(t, i, o, None, op)
            print(name, "done", flush=True)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    np.savez_compressed(os.path.join(HERE, "aniso_big.npz"), **out)
    print("aniso_big.npz", os.path.getsize(os.path.join(HERE, "aniso_big.npz")))


if __name__ == "__main__":
    main()
