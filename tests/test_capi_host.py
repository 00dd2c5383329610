"""CPU-side checks of the product: the C-ABI library loads and exports every symbol
include/djb_hip.h declares, host-only logic (microfacet::params resolution, file-name parsing,
sharding) matches the reference, and every compute entry point FAILS LOUDLY without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from dj_brdf_amd import _lib, djb, merl_params, shard, synth
from golden_cases import PARAM_CASES

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HAS_GPU = djb.device_count() > 0


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "djb_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(djb_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    syms = header_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(lib, s), f"libdjb_hip.so does not export {s}"
    assert sorted(_lib.EXPORTS) == syms, "dj_brdf_amd/_lib.py EXPORTS out of sync with include/djb_hip.h"
    assert lib.djb_version() == _lib.ABI_VERSION == 235      # include/djb_hip.h: the changelog of the ABI


# SURVEY.md section 8-N, measured with the reference compiled here: ggx isotropic(0.3), i = (0.3, 0.2, .), o = (-0.4, 0.1, .)
C_ABI_DEMO_KNOWN = ["eval 0.621380985 0.621380985 0.621380985", "pdf 0.581518769", "sample 0.657071352 0.080957301 0.749468625",
                    # a caller-defined BRDF (a C callback) fitted from samples at djb_fit_query_dirs == tabular(lambert) on the same context
                    "user-defined fit: 728 of 1023 query slots evaluated, alpha_ggx 0.693 (tabular(lambert): 0.693), tables identical",
                    # a caller-defined NDF (C callbacks restating GGX's radial functions) on the host path == the library's ggx
                    "user-defined NDF (GGX restated as callbacks): eval 0.621380985, identical to the library's ggx"]


def run_c_abi_demo(where):
    import subprocess
    inc = os.path.join(ROOT, "include")
    # the header is C: a cgo / JNI binding compiles against it without a C++ compiler
    src = os.path.join(ROOT, "examples", "c_abi_demo.c")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-I", inc, src], check=True)
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "examples"), "c_abi_demo"], check=True)
    r = subprocess.run([os.path.join(ROOT, "examples", "c_abi_demo"), where], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    return r.stdout.strip().splitlines()


def test_c_abi_from_plain_c_on_the_host_path():
    """examples/c_abi_demo.c -- C99, linked with gcc, no C++ on the caller's side -- reproduces the reference's known answers
    through the CPU context (the product's host path)."""
    out = run_c_abi_demo("cpu")
    assert out[0] == "device cpu" and out[1:] == C_ABI_DEMO_KNOWN, out


def test_params_resolve_matches_reference_golden():
    g = np.load(os.path.join(ROOT, "tests", "golden", "math.npz"))
    for k, p in enumerate(PARAM_CASES):
        if p is None:
            mp = djb.microfacet.params.standard()
        elif p[0] == "elliptic":
            mp = djb.microfacet.params.elliptic(*p[1:])
        else:
            mp = djb.microfacet.params.pdfparams(*p[1:])
        want = g[f"p{k}"]
        got = np.array(mp.get_location() + mp.get_ellipse() + mp.get_pdfparams() + (0.0,), np.float32)
        nan_ok = np.isnan(got) & np.isnan(want)
        assert np.array_equal(got.view(np.uint32)[~nan_ok], want.view(np.uint32)[~nan_ok]), (p, got, want)


def test_fresnel_ior_f0_helpers_match_reference_golden():
    # fresnel::ior_to_f0 / f0_to_ior (dj_brdf.h:151-154, 1255-1290): host-side helpers of the mirror
    g = np.load(os.path.join(ROOT, "tests", "golden", "math.npz"))
    got = djb.fresnel.ior_to_f0(g["ior_x"])
    assert np.array_equal(got.view(np.uint32), g["ior_f0"].view(np.uint32))
    got, want = djb.fresnel.f0_to_ior(g["f0_x"]), g["f0_ior"]
    ok = (got.view(np.uint32) == want.view(np.uint32)) | (np.isinf(got) & np.isinf(want) & (np.sign(got) == np.sign(want)))
    assert ok.all()
    got = djb.vec3_from_angles(g["ang_theta"], g["ang_phi"])                 # vec3(theta, phi), dj_brdf.h:589-595
    assert np.array_equal(got.view(np.uint32), g["ang_vec3"].view(np.uint32))
    assert djb.fresnel.ior_to_f0(1.5) == np.float32(0.2) * np.float32(0.2) and djb.fresnel.f0_to_ior(1.0) == 1.0


def test_invalid_params_raise_like_the_reference_asserts():
    for bad in (djb.microfacet.params.elliptic(0.0, 0.3), djb.microfacet.params.pdfparams(0.3, -1.0),
                djb.microfacet.params.pdfparams(0.3, 0.3, 1.0)):
        with pytest.raises(djb.exc):
            bad.get_ellipse()


@pytest.mark.skipif(HAS_GPU, reason="checks the no-GPU failure mode")
def test_no_gpu_means_loud_failure_not_fallback():
    assert djb.device_count() == 0
    with pytest.raises(djb.exc) as e:
        djb.Context(0)
    assert e.value.status_name == "DJB_ERR_NO_DEVICE"
    with pytest.raises(djb.exc):
        djb.ggx()
    with pytest.raises(djb.exc):
        merl_params.fit_files(["/tmp/x.binary"])


def test_null_arguments_are_rejected():
    lib = _lib.load()
    assert lib.djb_ctx_create(C.c_int(0), None) != 0
    assert lib.djb_device_count(None) != 0
    assert b"null" in lib.djb_last_error()
    assert lib.djb_params_resolve(None, None) != 0


def test_material_name_parsing():
    assert merl_params.material_name("/data/merl/gold-metallic-paint.binary") == "gold-metallic-paint"
    assert merl_params.material_name("x/a.b.c") == "a"
    assert merl_params.material_name("/p/" + "z" * 80 + ".binary") == "z" * 63
    txt = merl_params.format_params_txt(["/m/chrome.binary"], [(0.24549, 0.0606)])
    assert txt == "# MERL Beckmann GGX\nchrome 0.245 0.061\n"


def test_merl_payload_reader_errors(tmp_path):
    with pytest.raises(djb.exc) as e:
        merl_params.read_merl_payload(str(tmp_path / "missing.binary"))
    assert e.value.status_name == "DJB_ERR_OPEN_FAILED"
    bad = tmp_path / "bad.binary"; bad.write_bytes(np.array([0, 1, 2], np.int32).tobytes())
    with pytest.raises(djb.exc) as e:
        merl_params.read_merl_payload(str(bad))
    assert e.value.status_name == "DJB_ERR_BAD_HEADER"
    short = tmp_path / "short.binary"
    short.write_bytes(np.array([90, 90, 180], np.int32).tobytes() + b"\0" * 64)
    with pytest.raises(djb.exc) as e:
        merl_params.read_merl_payload(str(short))
    assert e.value.status_name == "DJB_ERR_READ_FAILED"


def test_sharding_covers_everything_once():
    for n in (0, 1, 7, 100, 1001):
        for w in (1, 2, 3, 8):
            blocks = [shard.block_range(n, w, r) for r in range(w)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[r][1] == blocks[r + 1][0] for r in range(w - 1))
            assert max(b[1] - b[0] for b in blocks) - min(b[1] - b[0] for b in blocks) <= 1
            rr = sorted(k for r in range(w) for k in shard.round_robin(n, w, r))
            assert rr == list(range(n))


def test_synth_file_format_and_determinism(tmp_path):
    t = synth.merl_table_hashed()
    p = tmp_path / "t.binary"
    synth.write_merl_binary(str(p), t)
    assert os.path.getsize(p) == synth.MERL_FILE_BYTES == 34992012
    back = merl_params.read_merl_payload(str(p))
    assert np.array_equal(back, t.reshape(-1))
    x, y, z = synth.directions(1000, 123, start=5)
    x2, y2, z2 = synth.directions(10, 123, start=500)
    assert np.array_equal(x[495:505], x2) and np.array_equal(z[495:505], z2)
    assert np.all(z > 0.04) and np.all(np.abs(x * x + y * y + z * z - 1) < 1e-6)
    u = synth.uniforms(100000, 9)
    assert 0 <= u.min() and u.max() < 1 and abs(u.mean() - 0.5) < 0.01
    assert len(set(synth.MERL_NAMES)) == 100


def test_params_that_carry_their_resolved_form():
    """include/djb_hip.h, ABI 231: a djb_params followed by what djb_params_resolve returned for it (kind | DJB_PARAMS_RESOLVED_FOLLOWS =
    djb_params_cached, the form of the facade's params objects) gives the bits of the plain djb_params -- one pair and a batch, every
    operator that takes parameters; djb_params_resolve itself ignores the flag (it is where a cached form comes from); a lambert brdf
    still refuses microfacet parameters whatever the flag says."""
    lib = _lib.load()
    FLAG = 0x100

    class Cached(C.Structure):
        _fields_ = [("p", _lib.Params), ("r", _lib.ParamsResolved)]

    ctx = djb.Context("cpu")
    g, bk, lam = djb.ggx(djb.fresnel.schlick((0.9, 0.6, 0.3)), True, ctx=ctx), djb.beckmann(ctx=ctx), djb.lambert(ctx=ctx)
    for n in (1, 257):
        i, o = synth.directions_aos(n, 5), synth.directions_aos(n, 6)
        u1, u2 = synth.uniforms(n, 7), synth.uniforms(n, 8)
        for case in PARAM_CASES:
            P = djb.microfacet.params
            p = getattr(P, case[0])(*case[1:]) if case is not None else P.standard()
            c = Cached(); c.p = p._p
            _lib.check(lib.djb_params_resolve(C.byref(c.p), C.byref(c.r)))
            r0 = _lib.ParamsResolved()
            c.p.kind |= FLAG
            poison = Cached(); C.memmove(C.byref(poison), C.byref(c), C.sizeof(c)); poison.r.ax = 123.0   # a stale resolved form next to a flagged kind
            _lib.check(lib.djb_params_resolve(C.byref(poison.p), C.byref(r0)))                        # ... is not what resolve reads
            assert bytes(r0) == bytes(c.r)

            Q = P.standard(); Q._p = c.p; Q._keep = c          # the mirror passes byref(_p): a view into `c`, so `r` follows it in memory
            for b in (g, bk):
                for op in ("eval", "evalp", "pdf"):
                    want, got = getattr(b, op)(i, o, p), getattr(b, op)(i, o, Q)
                    assert np.array_equal(np.asarray(want).view(np.uint32), np.asarray(got).view(np.uint32)), (case, op, n)
                want, got = b.sample(u1, u2, o, p), b.sample(u1, u2, o, Q)
                assert np.array_equal(want.view(np.uint32), got.view(np.uint32)), (case, "sample", n)
            if (c.p.kind & 0xff) == 0:                          # params::standard() is what a lambert accepts besides its own
                assert np.array_equal(lam.eval(i, o, p).view(np.uint32), lam.eval(i, o, Q).view(np.uint32))
            else:
                for q in (p, Q):
                    with pytest.raises(djb.exc):
                        lam.eval(i, o, q)
    ctx.close()
