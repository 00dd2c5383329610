#!/usr/bin/env python3
"""TEST INFRASTRUCTURE: drives the five Mitsuba BSDF shells through tests/mitsuba_mock/shell_harness.cpp.

Two sides share this file, the mock and the harness:
  * `--side ref`  : the REFERENCE's shells (/root/reference/mitsuba/*.cpp, unchanged) on /root/reference/dj_brdf.h,
                    built into oracle/_ref/shells/ -- build container only.  `--golden` stores the outputs as
                    tests/golden/shells.npz (data: inputs, outputs, queried property names, strings).
  * `--side refsrc`: the REFERENCE's shell sources, unchanged, compiled against THIS repository's facade (include/dj_brdf.h + libdjb_hip.so)
                    into oracle/_ref/shells_on_facade/ (built here, travels to the GPU box): the facade as a drop-in for the reference's own
                    callers, the sixth plugin dj_brdf.cpp included (`--which dj_brdf`, golden tests/golden/shells_dj_brdf.npz).
  * `--side repo` : this repository's shells (mitsuba/*.cpp) on include/djb_hip.hpp + libdjb_hip.so, built into
                    tests/mitsuba_mock/_build/.  DJB_DEVICE=cpu selects the library's host path, otherwise GPU 0;
                    DJB_SCALAR_ON_DEVICE=1 sends the one-hit calls through the kernels instead of the host twin.
tests/test_mitsuba_shells.py runs the repo side in a subprocess and compares with the golden file.

    python tests/mitsuba_mock/shell_cases.py --side ref --golden          # regenerate tests/golden/shells.npz
    python tests/mitsuba_mock/shell_cases.py --side repo --out /tmp/x.npz
"""
import argparse
import ctypes as C
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from dj_brdf_amd import synth  # noqa: E402

SHELLS = ["dj_merl", "dj_utia", "dj_abc", "dj_sgd", "dj_beckmannconductor"]
REF_DIR = os.path.join(ROOT, "oracle", "_ref", "shells")
REPO_DIR = os.path.join(HERE, "_build")
REFSRC_DIR = os.path.join(ROOT, "oracle", "_ref", "shells_on_facade")     # built from the reference's sources: lives with the other reference-built files

# BSDF::EBSDFType of the mock (mitsuba/mock.h)
DIFFUSE, GLOSSY, ALL = 0x2, 0x8, 0xFFFFFFFF
SOLID_ANGLE, DISCRETE = 1, 4
N = 192


def build(side, shells=SHELLS, quiet=True):
    """compile one shared object per shell; returns the directory"""
    cxx = shutil.which("g++") or shutil.which("c++")
    out = {"ref": REF_DIR, "refsrc": REFSRC_DIR}.get(side, REPO_DIR)
    os.makedirs(out, exist_ok=True)
    for s in shells:
        so = os.path.join(out, f"libshell_{s}.so")
        if side == "refsrc":
            # the REFERENCE's shell sources, unchanged, on this repository's facade: their `#include "dj_brdf.h"` finds include/dj_brdf.h
            src = f"/root/reference/mitsuba/{s}.cpp"
            inc = ["-I", os.path.join(ROOT, "include")]
            lib = os.path.join(ROOT, "dj_brdf_amd", "lib")
            link = ["-L", lib, "-ldjb_hip", f"-Wl,-rpath,{lib}"]
            deps = [src] + [os.path.join(ROOT, "include", f) for f in ("djb_hip.h", "djb_hip.hpp", "dj_brdf.h")]
            flags = ["-w"]
        elif side == "ref":
            src = f"/root/reference/mitsuba/{s}.cpp"
            inc = ["-I", "/root/reference"]
            link, deps = [], [src, "/root/reference/dj_brdf.h"]
            flags = ["-w"]
        else:
            src = os.path.join(ROOT, "mitsuba", f"{s}.cpp")
            inc = ["-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "mitsuba")]
            lib = os.path.join(ROOT, "dj_brdf_amd", "lib")
            link = ["-L", lib, "-ldjb_hip", f"-Wl,-rpath,{lib}"]
            deps = [src] + [os.path.join(ROOT, "mitsuba", f) for f in os.listdir(os.path.join(ROOT, "mitsuba")) if f.endswith(".hpp")] \
                + [os.path.join(ROOT, "include", f) for f in ("djb_hip.h", "djb_hip.hpp")]
            flags = ["-Wall", "-Wno-unused-parameter"]
        deps += [os.path.join(HERE, "shell_harness.cpp"), os.path.join(HERE, "mitsuba", "mock.h")]
        if os.path.exists(so) and all(os.path.getmtime(so) >= os.path.getmtime(d) for d in deps):
            continue
        cmd = [cxx, "-std=c++11", "-O2", "-fPIC", "-shared", "-DMITSUBA_MOCK_MAIN", "-DNVERBOSE",
               f'-DSHELL_SOURCE="{src}"', "-I", HERE] + inc + flags + [os.path.join(HERE, "shell_harness.cpp")] + link + ["-o", so]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"building {so} failed:\n{r.stderr[-4000:]}")
        if not quiet:
            print("built", so)
    return out


class Shell:
    """one loaded shell library"""
    def __init__(self, libdir, name):
        self.lib = C.CDLL(os.path.join(libdir, f"libshell_{name}.so"))
        self.lib.props_new.restype = C.c_void_p
        self.lib.shell_roughness.restype = C.c_float
        self.err = C.create_string_buffer(1024)

    def props(self, ident, **kv):
        p = C.c_void_p(self.lib.props_new(ident.encode()))
        for k, v in kv.items():
            if isinstance(v, bool):
                self.lib.props_set_boolean(p, k.encode(), C.c_int(int(v)))
            elif isinstance(v, str):
                self.lib.props_set_string(p, k.encode(), v.encode())
            elif isinstance(v, (tuple, list)):
                self.lib.props_set_spectrum(p, k.encode(), C.c_float(v[0]), C.c_float(v[1]), C.c_float(v[2]))
            else:
                self.lib.props_set_float(p, k.encode(), C.c_float(v))
        return p

    def _e(self, rc):
        return self.err.value.decode() if rc else ""

    def create(self, props):
        h = C.c_void_p()
        e = self._e(self.lib.shell_create(props, C.byref(h), self.err, 1024))
        return (h if not e else None), e

    def queried(self, props):
        b = C.create_string_buffer(4096)
        self.lib.props_queried(props, b, 4096)
        return sorted(set(b.value.decode().split("\n"))) if b.value else []

    def resolver_log(self):
        b = C.create_string_buffer(8192)
        self.lib.resolver_log(b, 8192)
        return b.value.decode()

    def add_child(self, h, name, coef=None):
        c = np.zeros(9, np.float32) if coef is None else np.asarray(coef, np.float32)
        return self._e(self.lib.shell_add_child(h, name.encode(), C.c_int(0 if coef is not None else 1),
                                                c.ctypes.data_as(C.c_void_p), self.err, 1024))

    def configure(self, h):
        return self._e(self.lib.shell_configure(h, self.err, 1024))

    def info(self, h):
        comp = (C.c_uint * 4)(); n = C.c_int(); rd = C.c_int()
        ens = C.create_string_buffer(1024); s = C.create_string_buffer(4096)
        e = self._e(self.lib.shell_info(h, comp, 4, C.byref(n), C.byref(rd), ens, 1024, s, 4096, self.err, 1024))
        assert not e, e
        return dict(components=np.array(list(comp)[:n.value], np.uint32), uses_rd=rd.value, ensured=ens.value.decode(),
                    string=s.value.decode())

    def roughness(self, h, u, v, comp=0):
        return float(self.lib.shell_roughness(h, C.c_float(u), C.c_float(v), C.c_int(comp)))

    def eval(self, h, wi, wo, uv, mask, comp, measure):
        n = len(wi); rgb = np.zeros((n, 3), np.float32); pdf = np.zeros(n, np.float32)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        e = self._e(self.lib.shell_eval(h, C.c_int(n), p(wi), p(wo), p(uv), p(mask), p(comp), p(measure), p(rgb), p(pdf), self.err, 1024))
        assert not e, e
        return rgb, pdf

    def sample(self, h, wi, uv, xi, mask, comp, with_pdf):
        n = len(wi); rgb = np.zeros((n, 3), np.float32); pdf = np.zeros(n, np.float32)
        wo = np.zeros((n, 3), np.float32); meta = np.zeros((n, 3), np.float32)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        e = self._e(self.lib.shell_sample(h, C.c_int(n), p(wi), p(uv), p(xi), p(mask), p(comp), C.c_int(int(with_pdf)),
                                          p(rgb), p(pdf), p(wo), p(meta), self.err, 1024))
        assert not e, e
        return rgb, pdf, wo, meta

    def serialize(self, h, rebuild):
        nb = C.c_int(); ni = C.c_int(); cc = C.c_uint(); s = C.create_string_buffer(4096)
        e = self._e(self.lib.shell_serialize(h, C.c_int(int(rebuild)), C.byref(nb), C.byref(ni), C.byref(cc), s, 4096, self.err, 1024))
        assert not e, e
        return nb.value, ni.value, cc.value, s.value.decode()

    def shader(self, h):
        code = C.create_string_buffer(16384); nd = C.c_int(); comp = C.c_int(); uni = C.create_string_buffer(1024)
        vals = np.zeros(9, np.float32); reg = C.c_int(); unreg = C.c_int()
        e = self._e(self.lib.shell_shader(h, code, 16384, C.byref(nd), C.byref(comp), uni, 1024, vals.ctypes.data_as(C.c_void_p),
                                          C.byref(reg), C.byref(unreg), self.err, 1024))
        assert not e, e
        return dict(code=code.value.decode(), n_deps=nd.value, complete=comp.value, uniforms=uni.value.decode(), values=vals,
                    registered=reg.value, unregistered=unreg.value)


def records():
    """the BSDFSamplingRecords every case is queried with: directions above AND below the horizon, all mask / component /
    measure combinations the shells' guards distinguish"""
    wi = synth.directions_aos(N, synth.SEED_I, start=70000)
    wo = synth.directions_aos(N, synth.SEED_O, start=70000)
    wi[5::16, 2] *= -1          # viewer below the surface
    wo[9::16, 2] *= -1          # light below the surface
    wi[13, 2] = 0.0; wo[29, 2] = 0.0                       # exactly on the horizon (<= 0 vs < 0 guards)
    uv = np.stack([synth.uniforms(N, 0xA11CE, start=1), synth.uniforms(N, 0xA11CE, start=5000)], axis=1).astype(np.float32)
    xi = np.stack([synth.uniforms(N, synth.SEED_U1, start=70000), synth.uniforms(N, synth.SEED_U2, start=70000)], axis=1).astype(np.float32)
    k = np.arange(N)
    mask = np.where(k % 4 == 1, DIFFUSE, np.where(k % 4 == 2, GLOSSY, ALL)).astype(np.uint32)
    comp = np.where(k % 7 == 3, 0, np.where(k % 7 == 5, 1, -1)).astype(np.int32)
    measure = np.where(k % 11 == 7, DISCRETE, SOLID_ANGLE).astype(np.int32)
    return dict(wi=np.ascontiguousarray(wi), wo=np.ascontiguousarray(wo), uv=uv, xi=xi, mask=mask, comp=comp, measure=measure)


def affine(a, bu=(0, 0, 0), cv=(0, 0, 0)):
    return list(a) + list(bu) + list(cv)


def cases(files):
    """(case name, shell, properties, children [(name, coef | None)]) -- every property / child the reference's shells consume"""
    merl, utia = files["merl"], files["utia"]
    lean1 = (25.03, 24.98, 0.0)               # texels carry the +25 bias on E1, E2 ...
    lean2 = (0.012, 0.02, 625.001)            # ... and +625 on E5
    c = [
        ("merl_default", "dj_merl", dict(filename=merl), []),
        ("merl_reflectance", "dj_merl", dict(filename=merl, reflectance=(0.2, 0.3, 0.4)), [("reflectance", affine((0.1, 0.2, 0.3)))]),
        ("merl_diffuse", "dj_merl", dict(filename=merl, diffuseReflectance=(0.6, 0.5, 0.4)), [("diffuseReflectance", affine((0.1, 0.2, 0.3)))]),
        ("merl_badchild", "dj_merl", dict(filename=merl), [("bump", affine((1, 1, 1)))]),
        ("merl_nontexture", "dj_merl", dict(filename=merl), [("reflectance", None)]),
        ("merl_nofile", "dj_merl", dict(), []),
        ("merl_missing", "dj_merl", dict(filename=os.path.join(files["dir"], "absent.binary")), []),
        ("utia_default", "dj_utia", dict(filename=utia), []),
        ("utia_reflectance", "dj_utia", dict(filename=utia, reflectance=(0.2, 0.3, 0.4)), [("reflectance", affine((0.1, 0.2, 0.3)))]),
        ("utia_badchild", "dj_utia", dict(filename=utia), [("alpha", affine((1, 1, 1)))]),
    ]
    for model in ("abc", "sgd"):
        c += [
            (f"{model}_gold", f"dj_{model}", dict(merlID="gold-metallic-paint"), []),
            (f"{model}_fabric", f"dj_{model}", dict(merlID="black-fabric", diffuseReflectance=(0.6, 0.5, 0.4)), []),
            (f"{model}_child", f"dj_{model}", dict(merlID="chrome", reflectance=(0.2, 0.3, 0.4)), [("reflectance", affine((0.1, 0.2, 0.3), (0.5, 0, 0)))]),
            (f"{model}_unknown", f"dj_{model}", dict(merlID="no-such-material"), []),
            (f"{model}_material_prop", f"dj_{model}", dict(material="chrome"), []),          # the wrong property name must fail
            (f"{model}_badchild", f"dj_{model}", dict(merlID="chrome"), [("specularReflectance", affine((1, 1, 1)))]),
        ]
    B = "dj_beckmannconductor"
    # the plugin subtracts the texel bias even without LEAN maps (constant-0 defaults -> E1 = E2 = -25: a lobe sheared out
    # of the hemisphere, bc_default); `flat` = maps of a flat surface, which leave the base lobe as it is
    flat = dict(leanmap1=(25.0, 25.0, 0.0), leanmap2=(0.0, 0.0, 625.0))
    c += [
        ("bc_default", B, dict(), []),
        ("bc_alpha", B, dict(alpha=0.3, **flat), []),
        ("bc_aniso", B, dict(alpha1=0.15, alpha2=0.4, alphaAngle=35.0, **flat), []),
        ("bc_lean", B, dict(alpha=0.2, leanmap1=lean1, leanmap2=lean2), []),
        ("bc_lean_scale_half", B, dict(alpha=0.2, leanmap1=lean1, leanmap2=lean2, dmapscale=0.5), []),
        ("bc_lean_scale_two", B, dict(alpha1=0.1, alpha2=0.3, alphaAngle=23.0, leanmap1=lean1, leanmap2=lean2, dmapscale=2.0), []),
        ("bc_lean_naive", B, dict(alpha=0.2, leanmap1=lean1, leanmap2=lean2, dmapscale=0.7, leanFiltering=False), []),
        ("bc_eta_k", B, dict(alpha=0.25, eta=(0.2, 0.9, 1.1), k=(3.9, 2.4, 2.2), specularReflectance=(0.9, 0.8, 0.7), **flat), []),
        ("bc_mitsuba_fresnel", B, dict(alpha=0.25, mitsubaFresnel=True, **flat), []),
        ("bc_material_au", B, dict(alpha=0.25, material="Au", extEta=1.33, **flat), []),
        ("bc_material_none_upper", B, dict(alpha=0.25, material="NONE", mitsubaFresnel=True, **flat), []),
        ("bc_merl", B, dict(merl=merl, alpha=0.8, **flat), []),
        ("bc_merl_default_alpha", B, dict(merl=merl, **flat), []),
        ("bc_merl_aniso", B, dict(merl=merl, alpha1=0.5, alpha2=1.5, alphaAngle=10.0, leanmap1=lean1, leanmap2=lean2), []),
        ("bc_textures", B, dict(alpha=0.3), [("alpha1", affine((0.15, 0.15, 0.15), (0.3, 0.3, 0.3))),
                                             ("alpha2", affine((0.4, 0.4, 0.4), (0, 0, 0), (-0.2, -0.2, -0.2))),
                                             ("alphaAngle", affine((0.1, 0.2, 0.3), (1.0, 1.0, 1.0))),
                                             ("leanmap1", affine(lean1, (0.05, -0.04, 0), (-0.03, 0.02, 0))),
                                             ("leanmap2", affine(lean2, (0.01, 0.0, 0.002), (0.0, 0.015, -0.001))),
                                             ("specularReflectance", affine((0.9, 0.8, 0.7), (0.05, 0.05, 0.05)))]),
        ("bc_alpha_child", B, dict(**flat), [("alpha", affine((0.35, 0.35, 0.35), (0.1, 0.1, 0.1)))]),
        ("bc_err_alpha_and_alpha1", B, dict(alpha=0.3, alpha1=0.2), []),
        ("bc_err_alpha_and_angle", B, dict(alpha=0.3, alphaAngle=20.0), []),
        ("bc_err_alpha1_only", B, dict(alpha1=0.3), []),
        ("bc_err_alpha2_only", B, dict(alpha2=0.3), []),
        ("bc_badchild", B, dict(), [("reflectance", affine((1, 1, 1)))]),
        ("bc_nontexture", B, dict(), [("alpha", None)]),
    ]
    return c


def cases_dj_brdf(files):
    """the sixth plugin, mitsuba/dj_brdf.cpp (not one of the five named ones; this repository ships no shell of its
    own for it).  Its REFERENCE source is compiled unchanged against the facade (side "refsrc") and must behave as on the reference's
    header: distribution beckmann / ggx / tabular, lobes fitted from a MERL or a UTIA file, both Fresnel modes."""
    merl, utia = files["merl"], files["utia"]
    D = "dj_brdf"
    c = [("db_default", D, dict(), [])]
    for dist in ("beckmann", "ggx", "GGX"):
        c += [(f"db_{dist}_alpha", D, dict(distribution=dist, alpha=0.3), []),
              (f"db_{dist}_aniso", D, dict(distribution=dist, alpha1=0.15, alpha2=0.4, alphaAngle=35.0), []),
              (f"db_{dist}_merl", D, dict(distribution=dist, merl=merl), []),
              (f"db_{dist}_merl_alpha", D, dict(distribution=dist, merl=merl, alpha=0.7, mitsubaFresnel=True), []),
              (f"db_{dist}_utia", D, dict(distribution=dist, utia=utia, alpha1=0.8, alpha2=1.2, alphaAngle=15.0), [])]
    c += [
        ("db_tabular_merl", D, dict(distribution="tabular", merl=merl), []),
        ("db_tabular_merl_mf", D, dict(distribution="tabular", merl=merl, mitsubaFresnel=True, material="Au", alpha=0.9), []),
        ("db_tabular_utia", D, dict(distribution="tabular", utia=utia), []),
        ("db_tabular_nofile", D, dict(distribution="tabular"), []),
        ("db_eta_k", D, dict(alpha=0.25, eta=(0.2, 0.9, 1.1), k=(3.9, 2.4, 2.2), specularReflectance=(0.9, 0.8, 0.7)), []),
        ("db_material_au", D, dict(alpha=0.25, material="Au", extEta=1.33), []),
        ("db_textures", D, dict(alpha=0.3), [("alpha1", affine((0.15, 0.15, 0.15), (0.3, 0.3, 0.3))), ("alpha2", affine((0.4, 0.4, 0.4), (0, 0, 0), (-0.2, -0.2, -0.2))),
                                             ("alphaAngle", affine((0.1, 0.2, 0.3), (1.0, 1.0, 1.0))), ("specularReflectance", affine((0.9, 0.8, 0.7), (0.05, 0.05, 0.05)))]),
        ("db_alpha_child", D, dict(), [("alpha", affine((0.35, 0.35, 0.35), (0.1, 0.1, 0.1)))]),
        ("db_err_distribution", D, dict(distribution="phong"), []),
        ("db_err_alpha_and_alpha1", D, dict(alpha=0.3, alpha1=0.2), []),
        ("db_err_alpha1_only", D, dict(alpha1=0.3), []),
        ("db_err_missing_merl", D, dict(merl=os.path.join(files["dir"], "absent.binary")), []),
        ("db_badchild", D, dict(), [("reflectance", affine((1, 1, 1)))]),
        ("db_nontexture", D, dict(), [("alpha", None)]),
    ]
    return c


def write_inputs(d):
    merl = os.path.join(d, "shells-material.binary")
    synth.write_merl_binary(merl, synth.merl_table(0.25, (0.10, 0.08, 0.05), (0.9, 0.7, 0.4)))
    utia = os.path.join(d, "shells-material.bin")
    synth.utia_table_smooth().astype(np.float64).tofile(utia)
    return dict(dir=d, merl=merl, utia=utia)


def run(side, libdir=None, which="five"):
    """which: "five" = the five named plugins (cases), "dj_brdf" = the sixth (cases_dj_brdf)"""
    libdir = libdir or build(side, SHELLS if which == "five" else ["dj_brdf"])
    out = {}
    tmp = tempfile.mkdtemp(prefix="djb_shells_")
    try:
        files = write_inputs(tmp)
        rec = records()
        for k, v in rec.items():
            out[f"in_{k}"] = v
        loaded = {}
        for name, shell, props_kv, children in (cases(files) if which == "five" else cases_dj_brdf(files)):
            S = loaded.get(shell) or loaded.setdefault(shell, Shell(libdir, shell))
            S.resolver_log()
            props = S.props(name, **props_kv)
            h, err = S.create(props)
            strs = {"create_error": err.replace(tmp, "<dir>")}
            strs["queried"] = "\n".join(S.queried(props))
            strs["resolved"] = S.resolver_log().replace(tmp, "<dir>")
            if h is not None:
                strs["child_errors"] = "\n".join(f"{n}: {S.add_child(h, n, coef)}" for n, coef in children)
                strs["configure_error"] = S.configure(h)
                info = S.info(h)
                out[f"{name}/components"] = info["components"]
                out[f"{name}/uses_rd"] = np.array([info["uses_rd"]], np.int32)
                strs["ensured"] = info["ensured"]; strs["toString"] = info["string"]
                out[f"{name}/roughness"] = np.array([S.roughness(h, 0.25, 0.75), S.roughness(h, 0.9, 0.1)], np.float32)
                rgb, pdf = S.eval(h, rec["wi"], rec["wo"], rec["uv"], rec["mask"], rec["comp"], rec["measure"])
                out[f"{name}/eval"], out[f"{name}/pdf"] = rgb, pdf
                for with_pdf in (0, 1):
                    rgb, pdf, wo, meta = S.sample(h, rec["wi"], rec["uv"], rec["xi"], rec["mask"], rec["comp"], with_pdf)
                    t = f"{name}/sample{2 + with_pdf}"
                    out[f"{t}_value"], out[f"{t}_pdf"], out[f"{t}_wo"], out[f"{t}_meta"] = rgb, pdf, wo, meta
                # the unserializing constructors of the measured-material shells never set their djb pointers: serialize only;
                # dj_beckmannconductor / dj_sgd restore their textures: rebuild (the clone is leaked, see the harness)
                rebuild = shell in ("dj_beckmannconductor", "dj_sgd")
                nb, ni, cc, cs = S.serialize(h, rebuild)
                out[f"{name}/serialized"] = np.array([nb - len(name), ni, cc], np.int64)
                strs["clone"] = cs
                sh = S.shader(h)
                strs["shader_code"] = sh["code"]; strs["shader_uniforms"] = sh["uniforms"]
                out[f"{name}/shader"] = np.array([sh["n_deps"], sh["complete"], sh["registered"], sh["unregistered"]], np.int32)
                out[f"{name}/shader_values"] = sh["values"]
            for k, v in strs.items():
                out[f"{name}/str_{k}"] = np.array(v)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--side", choices=["ref", "repo", "refsrc"], required=True)
    ap.add_argument("--out")
    ap.add_argument("--golden", action="store_true")
    ap.add_argument("--which", choices=["five", "dj_brdf"], default="five")
    ap.add_argument("--prebuilt", action="store_true", help="use the libraries already in the side's directory (the GPU box has no /root/reference to build refsrc from)")
    a = ap.parse_args()
    res = run(a.side, libdir={"ref": REF_DIR, "refsrc": REFSRC_DIR, "repo": REPO_DIR}[a.side] if a.prebuilt else None, which=a.which)
    path = os.path.join(ROOT, "tests", "golden", "shells.npz" if a.which == "five" else "shells_dj_brdf.npz") if a.golden else a.out
    if a.golden:
        assert a.side == "ref", "the golden file comes from the reference's shells"
    np.savez_compressed(path, **res)
    print(path, os.path.getsize(path), "bytes,", len(res), "arrays")


if __name__ == "__main__":
    main()
