#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REAL reference.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

It loads oracle/_ref/libdjb_ref.so -- the unmodified /root/reference/dj_brdf.h compiled in place
behind oracle/ref_shim.cpp -- and oracle/_ref/merl_params (the reference's own example driver),
feeds them the seeded synthetic inputs of dj_brdf_amd/synth.py, and stores inputs + outputs as
small .npz / .txt fixtures.  The fixtures are data; no reference source is stored.
"""
import hashlib
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oraclelib  # noqa: E402
from dj_brdf_amd import synth  # noqa: E402
from golden_cases import (FIT_CASES, MICROFACET_CASES, N_FIT_EVAL, N_HD, N_MERL, N_MICROFACET,  # noqa: E402
                          PARAM_CASES, PARAMS_TXT_MATERIALS)


ONLY = set(sys.argv[1:])      # e.g. `make_golden.py lean.npz`: rewrite just that fixture (the others stay byte-identical in git)


def save(name, **arrays):
    if not ONLY or name in ONLY:
        np.savez_compressed(os.path.join(HERE, name), **arrays)


def main():
    R = oraclelib.reference()
    assert R is not None, "needs /root/reference (build container)"

    # ---- microfacet eval / evalp / pdf / sample / evalp_is
    i = synth.directions_aos(N_MICROFACET, synth.SEED_I)
    o = synth.directions_aos(N_MICROFACET, synth.SEED_O)
    u1 = synth.uniforms(N_MICROFACET, synth.SEED_U1)
    u2 = synth.uniforms(N_MICROFACET, synth.SEED_U2)
    out = {"i": i, "o": o, "u1": u1, "u2": u2}
    for k, (ndf, fres, shadow, par) in enumerate(MICROFACET_CASES):
        b = R.microfacet(ndf, fres, shadow)
        for op in ("eval", "evalp", "pdf"):
            out[f"c{k}_{op}"] = R.eval(b, i, o, par, op)
        out[f"c{k}_sample"] = R.sample(b, u1, u2, o, par)
        w, si, pdf = R.evalp_is(b, u1, u2, o, par)
        out[f"c{k}_is_w"], out[f"c{k}_is_i"], out[f"c{k}_is_pdf"] = w, si, pdf
    save("microfacet.npz", **out)

    # ---- params, special functions, half/diff transforms
    out = {}
    for k, p in enumerate(PARAM_CASES):
        out[f"p{k}"] = R.params_get(p)
    x = np.linspace(-4, 4, 4001).astype(np.float32)
    out["erf_x"], out["erf_y"] = x, R.erf(x)
    x = np.linspace(-0.99999, 0.99999, 4001).astype(np.float32)
    out["erfinv_x"], out["erfinv_y"] = x, R.erfinv(x)
    th = np.linspace(0.0, np.pi, 1501).astype(np.float32); ph = np.linspace(-7.0, 7.0, 1501).astype(np.float32)
    out["ang_theta"], out["ang_phi"], out["ang_vec3"] = th, ph, R.vec3_angles(th, ph)   # vec3(theta, phi)
    x = np.concatenate([np.linspace(1.0, 6.0, 2001), [0.25, 0.5, 1.33, 1.5]]).astype(np.float32)
    out["ior_x"], out["ior_f0"] = x, R.ior_f0(0, x)                     # fresnel::ior_to_f0
    x = np.concatenate([np.linspace(0.0, 1.0, 2001), [0.04, 0.9999999]]).astype(np.float32)
    out["f0_x"], out["f0_ior"] = x, R.ior_f0(1, x)                      # fresnel::f0_to_ior
    i = synth.directions_aos(N_HD, synth.SEED_I, start=1000)
    o = synth.directions_aos(N_HD, synth.SEED_O, start=1000)
    h, d = R.io_to_hd(i, o)
    bi, bo = R.hd_to_io(h, d)
    out.update(hd_i=i, hd_o=o, hd_h=h, hd_d=d, hd_back_i=bi, hd_back_o=bo)
    save("math.npz", **out)

    # ---- sgd / abc analytic models (published parameter tables) and the fit the dj_abc / dj_sgd
    #      plugins run at load time: tabular(model, 90)  (mitsuba/dj_abc.cpp:31, dj_sgd.cpp:31)
    from golden_cases import MODEL_MATERIALS, N_MODEL
    i = synth.directions_aos(N_MODEL, synth.SEED_I, start=20000)
    o = synth.directions_aos(N_MODEL, synth.SEED_O, start=20000)
    out = {"i": i, "o": o}
    for kind in ("sgd", "abc"):
        for name in MODEL_MATERIALS:
            b = getattr(R, kind)(name)
            out[f"{kind}_{name}_eval"] = R.eval(b, i, o)
            out[f"{kind}_{name}_evalp"] = R.eval(b, i, o, None, "evalp")
            # member queries sgd::{ndf,gaf,g1,fresnel} / abc::{ndf,gaf,fresnel}: h := i, (i, o) := (o, i)
            cc = np.zeros_like(i); cc[:, 0] = np.clip(i[:, 2], 0, 1)
            out[f"{kind}_{name}_ndf"] = R.model_query(b, "ndf", i)
            out[f"{kind}_{name}_gaf"] = R.model_query(b, "gaf", i, o, i)
            out[f"{kind}_{name}_fresnel"] = R.model_query(b, "fresnel", cc)
            if kind == "sgd":
                out[f"{kind}_{name}_g1"] = R.model_query(b, "g1", o)
        t = R.tabular(getattr(R, kind)(MODEL_MATERIALS[0]), 90, True)
        for k, v in R.tabular_tables(t).items():
            out[f"{kind}_fit_{k}"] = np.atleast_1d(v)
    save("models.npz", **out)

    # ---- beckmann::lrep algebra and the per-hit LEAN path of dj_beckmannconductor
    from golden_cases import LEAN_BASE, LEAN_CASES, N_LEAN, lean_moments, lean_texels, lrep_cases
    out = {}
    for k, (op, a, b, x, y) in enumerate(lrep_cases()):
        out[f"lrep{k}"] = R.lrep_op(op, a, b, x, y)
    for k, p in enumerate(PARAM_CASES):
        out[f"roundtrip{k}"] = R.params_lrep_roundtrip(p)
    i = synth.directions_aos(N_LEAN, synth.SEED_I, start=30000)
    o = synth.directions_aos(N_LEAN, synth.SEED_O, start=30000)
    lean = lean_moments(N_LEAN)
    u1 = synth.uniforms(N_LEAN, synth.SEED_U1, start=30000)
    u2 = synth.uniforms(N_LEAN, synth.SEED_U2, start=30000)
    out.update(i=i, o=o, lean=lean, u1=u1, u2=u2)
    for c, (scale, filtering, biased) in enumerate(LEAN_CASES):
        tex = lean_texels(lean, biased)
        for ndf in ("beckmann", "ggx"):
            b = R.microfacet(ndf, ("schlick", 1.0, 0.71, 0.29), True)
            for op in ("eval", "evalp", "pdf"):
                val, pp = R.eval_lean(b, i, o, LEAN_BASE, scale, tex, op, filtering=filtering, biased=biased)
                out[f"c{c}_{ndf}_{op}"] = val
            out[f"c{c}_pdfparams"] = pp
            # dj_beckmann_conductor::sample per hit: evalp_is (and sample) with the per-hit params
            w, si, pdf, pp2 = R.sample_lean(b, u1, u2, o, LEAN_BASE, scale, tex, True, filtering=filtering, biased=biased)
            assert np.array_equal(pp2.view(np.uint32), pp.view(np.uint32))
            out[f"c{c}_{ndf}_is_w"], out[f"c{c}_{ndf}_is_i"], out[f"c{c}_{ndf}_is_pdf"] = w, si, pdf
            out[f"c{c}_{ndf}_sample"] = R.sample_lean(b, u1, u2, o, LEAN_BASE, scale, tex, False, filtering=filtering, biased=biased)[0]
    save("lean.npz", **out)

    # ---- tabular_anisotropic: fit tables, moment fits, sampling queries, operators
    from golden_cases import ANISO_CASES, N_ANISO, aniso_source
    i = synth.directions_aos(N_ANISO, synth.SEED_I, start=40000)
    o = synth.directions_aos(N_ANISO, synth.SEED_O, start=40000)
    u1 = synth.uniforms(N_ANISO, synth.SEED_U1, start=40000)
    u2 = synth.uniforms(N_ANISO, synth.SEED_U2, start=40000)
    out = {"i": i, "o": o, "u1": u1, "u2": u2}
    tmpa = tempfile.mkdtemp(prefix="djb_golden_aniso_")
    for name, (src, elev, azim, shadow) in ANISO_CASES.items():
        t = R.tabular_anisotropic(aniso_source(R, src, tmpa), elev, azim, shadow)
        for k, v in R.aniso_tables(t).items():
            out[f"{name}_{k}"] = v
        phi, th = (u1 * np.float32(6.2)).astype(np.float32), (u2 * np.float32(1.5)).astype(np.float32)
        for q, args in (("pdf1", (phi,)), ("cdf1", (phi,)), ("qf1", (u1,)), ("pdf2", (th, phi)),
                        ("cdf2", (th, phi)), ("qf2", (u2, phi))):
            out[f"{name}_{q}"] = R.aniso_query(t, q, *args)
        for op in ("eval", "evalp", "pdf"):
            out[f"{name}_{op}"] = R.eval(t, i, o, None, op)
        out[f"{name}_eval_ell"] = R.eval(t, i, o, ("elliptic", 0.2, 0.5, 0.7), "eval")
        out[f"{name}_sample"] = R.sample(t, u1, u2, o)
    shutil.rmtree(tmpa, ignore_errors=True)
    save("aniso.npz", **out)

    # ---- user-defined classes: a BRDF derived from djb::brdf, a Fresnel term derived from djb::fresnel::impl
    from golden_cases import (CUSTOM_ANISO, CUSTOM_FITS, CUSTOM_FRESNEL, CUSTOM_FRESNEL_FIT, CUSTOM_LOBES, CUSTOM_PARAMS,
                              N_CUSTOM)
    i = synth.directions_aos(N_CUSTOM, synth.SEED_I, start=50000)
    o = synth.directions_aos(N_CUSTOM, synth.SEED_O, start=50000)
    u1 = synth.uniforms(N_CUSTOM, synth.SEED_U1, start=50000)
    u2 = synth.uniforms(N_CUSTOM, synth.SEED_U2, start=50000)
    h, d = R.io_to_hd(i, o)
    out = {"i": i, "o": o, "u1": u1, "u2": u2, "h": h, "d": d}
    for name, lobe in CUSTOM_LOBES.items():
        b = R.custom(*lobe)
        for op in ("eval", "evalp", "pdf"):
            out[f"{name}_{op}"] = R.eval(b, i, o, None, op)
        for op in ("eval_hd", "evalp_hd"):
            out[f"{name}_{op}"] = R.eval(b, h, d, None, op)
        out[f"{name}_sample"] = R.sample(b, u1, u2, o)
        out[f"{name}_is_w"], out[f"{name}_is_i"], out[f"{name}_is_pdf"] = R.evalp_is(b, u1, u2, o)
        for res, shadow in CUSTOM_FITS:
            for k, v in R.tabular_tables(R.tabular(b, res, shadow)).items():
                out[f"{name}_fit{res}_{k}"] = np.atleast_1d(v)
        t = R.tabular_anisotropic(b, *CUSTOM_ANISO)
        for k, v in R.aniso_tables(t).items():
            out[f"{name}_aniso_{k}"] = v
    for ndf in ("ggx", "beckmann"):
        for shadow in (True, False):
            b = R.microfacet(ndf, CUSTOM_FRESNEL, shadow)
            tag = f"{ndf}{int(shadow)}"
            for op in ("eval", "evalp"):
                out[f"{tag}_{op}"] = R.eval(b, i, o, CUSTOM_PARAMS, op)
            for op in ("eval_hd", "evalp_hd"):
                out[f"{tag}_{op}"] = R.eval(b, h, d, CUSTOM_PARAMS, op)
            out[f"{tag}_is_w"], out[f"{tag}_is_i"], out[f"{tag}_is_pdf"] = R.evalp_is(b, u1, u2, o, CUSTOM_PARAMS)
            out[f"{tag}_fresnel"] = R.fresnel_eval(b, np.clip(o[:, 2], 0, 1))
        # eval_hd / evalp_hd of the library's own classes too (evalp_hd is eval * cos, hdr:808-814, also where evalp is overridden)
        b = R.microfacet(ndf, ("schlick", 0.9, 0.5, 0.1), True)
        for op in ("eval_hd", "evalp_hd"):
            out[f"{ndf}_schlick_{op}"] = R.eval(b, h, d, CUSTOM_PARAMS, op)
        t = R.tabular(R.microfacet(ndf, CUSTOM_FRESNEL, True), CUSTOM_FRESNEL_FIT, True)
        for k, v in R.tabular_tables(t).items():
            out[f"{ndf}_fit_{k}"] = np.atleast_1d(v)
    # user-defined NDFs: classes derived from djb::radial / djb::microfacet, with a library Fresnel term and with the user's
    from golden_cases import CUSTOM_NDFS, CUSTOM_NDF_PARAMS, CUSTOM_NDF_QUERIES
    M = 128      # the first M pairs: every element is one scalar call of a user virtual on the other side
    i, o, u1, u2, h, d = i[:M], o[:M], u1[:M], u2[:M], h[:M], d[:M]
    for ndf in CUSTOM_NDFS:
        for fk, fres in (("ideal", ("ideal",)), ("schlick", ("schlick", 0.9, 0.5, 0.1)), ("user", CUSTOM_FRESNEL)):
            for shadow in (True, False):
                b = R.microfacet(ndf, fres, shadow)
                for pk, par in enumerate(CUSTOM_NDF_PARAMS):
                    tag = f"{ndf}_{fk}{int(shadow)}_p{pk}"
                    for op in ("eval", "evalp", "pdf"):
                        out[f"{tag}_{op}"] = R.eval(b, i, o, par, op)
                    out[f"{tag}_evalp_hd"] = R.eval(b, h, d, par, "evalp_hd")
                    out[f"{tag}_sample"] = R.sample(b, u1, u2, o, par)
                    out[f"{tag}_is_w"], out[f"{tag}_is_i"], out[f"{tag}_is_pdf"] = R.evalp_is(b, u1, u2, o, par)
                    if fk == "ideal" and shadow:
                        args = {"h": i, "i": i, "o": o}
                        for q, sig in CUSTOM_NDF_QUERIES:
                            out[f"{tag}_{q}"] = R.microfacet_query(b, q, *[args[c] for c in sig], params=par)
        b = R.microfacet(ndf, ("ideal",), True)
        for k, v in R.tabular_tables(R.tabular(b, 40, True)).items():
            out[f"{ndf}_fit_{k}"] = np.atleast_1d(v)
        for k, v in R.aniso_tables(R.tabular_anisotropic(b, *CUSTOM_ANISO)).items():
            out[f"{ndf}_aniso_{k}"] = v
    save("custom.npz", **out)

    # ---- MERL lookup (hash-filled table: exact on any machine)
    tmp = tempfile.mkdtemp(prefix="djb_golden_")
    try:
        i = synth.directions_aos(N_MERL, synth.SEED_I, start=5000)
        o = synth.directions_aos(N_MERL, synth.SEED_O, start=5000)
        path = os.path.join(tmp, "hashed.binary")
        synth.write_merl_binary(path, synth.merl_table_hashed())
        m = R.merl(path)
        save("merl.npz", i=i, o=o, index=R.merl_index(i, o), eval=R.eval(m, i, o),
             evalp=R.eval(m, i, o, None, "evalp"), pdf=R.eval(m, i, o, None, "pdf"))
        if ONLY and not ({"fit.npz", "params_expected.txt"} & ONLY):
            return

        # ---- the fitter
        i = synth.directions_aos(N_FIT_EVAL, synth.SEED_I, start=9000)
        o = synth.directions_aos(N_FIT_EVAL, synth.SEED_O, start=9000)
        u1 = synth.uniforms(N_FIT_EVAL, synth.SEED_U1, start=9000)
        u2 = synth.uniforms(N_FIT_EVAL, synth.SEED_U2, start=9000)
        out = {"i": i, "o": o, "u1": u1, "u2": u2}
        for name, (src, res, shadow) in FIT_CASES.items():
            if src[0] == "merl":
                tab = synth.merl_table(*src[1:])
                out[f"{name}_table_sha256"] = np.frombuffer(hashlib.sha256(tab.tobytes()).digest(), np.uint8)
                path = os.path.join(tmp, f"{name}.binary")
                synth.write_merl_binary(path, tab)
                s = R.merl(path)
            else:
                s = R.microfacet(src[0], ("ideal",), src[1])
            t = R.tabular(s, res, shadow)
            for k, v in R.tabular_tables(t).items():
                out[f"{name}_{k}"] = np.atleast_1d(v)
            out[f"{name}_eval"] = R.eval(t, i, o, None, "eval")
            out[f"{name}_pdf"] = R.eval(t, i, o, None, "pdf")
            out[f"{name}_sample"] = R.sample(t, u1, u2, o)
        save("fit.npz", **out)

        # ---- params.txt of the reference's own example driver (examples/merl_params.cpp)
        exe = oraclelib.ref_merl_params_binary()
        files = []
        for name, recipe in PARAMS_TXT_MATERIALS:
            path = os.path.join(tmp, name + ".binary")
            synth.write_merl_binary(path, synth.merl_table(*recipe))
            files.append(path)
        subprocess.run([exe] + files, cwd=tmp, check=True, stdout=subprocess.DEVNULL)
        if not ONLY or "params_expected.txt" in ONLY:
            shutil.copy(os.path.join(tmp, "params.txt"), os.path.join(HERE, "params_expected.txt"))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    for f in sorted(os.listdir(HERE)):
        print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
