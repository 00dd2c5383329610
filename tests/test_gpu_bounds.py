"""Out-of-bounds writes and unwritten outputs: every batch operator of the C ABI on device memory with awkward sizes
(1, 3, 63, 65, 255, 257, 1023, 4097, 100003 units: below / across the 4-pair, wave, workgroup and chunk granularities of the
kernels), outputs placed between sentinel bands in a larger allocation -- SoA planes and AoS records, 16-byte-aligned and
misaligned starts -- in both numerical modes.  After each call the bands must be untouched and every output element written
(no sentinel left inside), and the inputs unchanged."""
import ctypes as C

import numpy as np
import pytest

from dj_brdf_amd import djb, synth, _lib

pytestmark = pytest.mark.gpu

PAD = 67                                   # odd on purpose: the payload starts 4-byte-aligned only
SENT = np.float32(np.frombuffer(np.uint32(0x7FC0DEAD).tobytes(), np.float32)[0])
SENT_BITS = 0x7FC0DEAD
SIZES = (1, 3, 63, 65, 255, 257, 1023, 4097, 100003)


class Buf:
    """`planes` float planes of n elements each, every plane between two sentinel bands; aos: one band pair around n records of 3"""

    def __init__(self, torch, dev, n, planes=3, aos=False, fill=None, align16=False):
        self.n, self.planes, self.aos = n, planes, aos
        self.pad = 64 if align16 else PAD
        self.row = n + 2 * self.pad
        self.t = torch.full((planes * self.row,), float("nan"), dtype=torch.float32, device=dev)
        self.t.view(torch.int32).fill_(SENT_BITS)
        if fill is not None:
            for k in range(planes):
                self.plane(k).copy_(torch.as_tensor(np.ascontiguousarray(fill[k]), device=dev))
        self.torch = torch

    def plane(self, k):
        if self.aos:
            return self.t[3 * self.pad + k: 3 * self.pad + 3 * self.n: 3]
        return self.t[k * self.row + self.pad: k * self.row + self.pad + self.n]

    def ptr(self, k=0):
        return self.t.data_ptr() + 4 * ((3 * self.pad + k) if self.aos else (k * self.row + self.pad))

    def view(self):
        v = _lib.Vec3View()
        v.x, v.y, v.z, v.stride = self.ptr(0), self.ptr(1), self.ptr(2), 3 if self.aos else 1
        return v

    def check(self, what, written=True):
        bits = self.t.view(self.torch.int32)
        inside = self.torch.zeros_like(bits, dtype=self.torch.bool)
        if self.aos:
            inside[3 * self.pad: 3 * self.pad + 3 * self.n] = True
        else:
            for k in range(self.planes):
                inside[k * self.row + self.pad: k * self.row + self.pad + self.n] = True
        sent = bits == (SENT_BITS)
        assert bool(sent[~inside].all()), f"{what}: a sentinel band was written ({int((~sent[~inside]).sum())} elements)"
        if written:
            assert not bool(sent[inside].any()), f"{what}: {int(sent[inside].sum())} output elements were never written"


def _objects(ctx):
    merl = djb.merl.from_table(synth.merl_table(), ctx=ctx)
    utia = djb.utia.from_table(synth.utia_table_smooth(), ctx=ctx)
    return {
        "ggx": djb.ggx(djb.fresnel.schlick((1.0, 0.71, 0.29)), True, ctx=ctx),
        "beckmann": djb.beckmann(ctx=ctx),
        "beckmann_sharp": djb.beckmann(djb.fresnel.unpolarized((1.5, 1.8, 2.4)), False, ctx=ctx),
        "merl": merl, "utia": utia, "lambert": djb.lambert(ctx=ctx),
        "sgd": djb.sgd("gold-metallic-paint", ctx=ctx), "abc": djb.abc("chrome", ctx=ctx),
        "tabular": djb.tabular(merl, 90, True, ctx=ctx),
        "aniso": djb.tabular_anisotropic(utia, 10, 14, True, ctx=ctx),
    }


@pytest.mark.parametrize("contract", [False, True])
@pytest.mark.parametrize("layout", ["soa", "soa16", "aos"])
def test_batch_operators_stay_inside_their_outputs(gpu_ctx, contract, layout):
    import torch
    lib = _lib.load()
    dev = f"cuda:{gpu_ctx.device}"
    objs = _objects(gpu_ctx)
    P = djb.microfacet.params
    params = {"ggx": P.elliptic(0.2, 0.5, 0.7), "beckmann": P.elliptic(0.2, 0.5, 0.7), "beckmann_sharp": P.isotropic(0.05),
              "tabular": P.isotropic(0.4), "aniso": None}
    aos, a16 = layout == "aos", layout == "soa16"
    djb.set_contract_1e5(gpu_ctx, contract)
    try:
        for n in SIZES:
            hi, ho = synth.directions(n, 11), synth.directions(n, 12)          # [3, n]
            u1, u2 = synth.uniforms(n, 13), synth.uniforms(n, 14)
            mk = lambda planes=3, fill=None, a=aos: Buf(torch, dev, n, planes, a and planes == 3, fill, a16)
            for name, b in objs.items():
                p = params.get(name)
                pp = C.byref(p._p) if p is not None else None
                bi, bo = mk(fill=hi), mk(fill=ho)
                bu1, bu2 = mk(1, [u1]), mk(1, [u2])
                vi, vo = bi.view(), bo.view()
                h, cx = b._h, gpu_ctx._h
                tag = f"{name} n={n} {layout} contract={contract}"
                out, pdf = mk(), mk(1)
                vout = out.view()
                _lib.check(lib.djb_eval_batch(cx, h, C.c_int64(n), C.byref(vi), C.byref(vo), pp, C.byref(vout), C.c_int(_lib.MEM_DEVICE)))
                torch.cuda.synchronize(); out.check(tag + " eval")
                out = mk(); vout = out.view()
                _lib.check(lib.djb_evalp_batch(cx, h, C.c_int64(n), C.byref(vi), C.byref(vo), pp, C.byref(vout), C.c_int(_lib.MEM_DEVICE)))
                torch.cuda.synchronize(); out.check(tag + " evalp")
                _lib.check(lib.djb_pdf_batch(cx, h, C.c_int64(n), C.byref(vi), C.byref(vo), pp, C.c_void_p(pdf.ptr()), C.c_int(_lib.MEM_DEVICE)))
                torch.cuda.synchronize(); pdf.check(tag + " pdf")
                for cos in (0, 1):
                    out, pdf = mk(), mk(1); vout = out.view()
                    _lib.check(lib.djb_eval_pdf_batch(cx, h, C.c_int64(n), C.byref(vi), C.byref(vo), pp, C.c_int(cos), C.byref(vout), C.c_void_p(pdf.ptr()),
                                                      C.c_int(_lib.MEM_DEVICE)))
                    torch.cuda.synchronize(); out.check(tag + f" eval_pdf cos={cos}"); pdf.check(tag + f" eval_pdf cos={cos} (pdf)")
                oi = mk(); voi = oi.view()
                _lib.check(lib.djb_sample_batch(cx, h, C.c_int64(n), C.c_void_p(bu1.ptr()), C.c_void_p(bu2.ptr()), C.byref(vo), pp, C.byref(voi), C.c_int(_lib.MEM_DEVICE)))
                torch.cuda.synchronize(); oi.check(tag + " sample")
                oi = mk(); voi = oi.view()
                _lib.check(lib.djb_sample_rng_batch(cx, h, C.c_int64(n), C.c_uint32(5), C.c_uint32(6), C.c_uint64(17), C.byref(vo), pp, C.byref(voi)))
                torch.cuda.synchronize(); oi.check(tag + " sample_rng")
                ow, oi, pdf = mk(), mk(), mk(1); vow, voi = ow.view(), oi.view()
                _lib.check(lib.djb_evalp_is_batch(cx, h, C.c_int64(n), C.c_void_p(bu1.ptr()), C.c_void_p(bu2.ptr()), C.byref(vo), pp, C.byref(vow), C.byref(voi),
                                                  C.c_void_p(pdf.ptr()), C.c_int(_lib.MEM_DEVICE)))
                torch.cuda.synchronize(); ow.check(tag + " evalp_is (weight)"); oi.check(tag + " evalp_is (i)"); pdf.check(tag + " evalp_is (pdf)")
                for inp, what in ((bi, "i"), (bo, "o"), (bu1, "u1"), (bu2, "u2")):
                    inp.check(tag + f" input {what}", written=False)
                assert np.array_equal(bi.plane(2).cpu().numpy(), hi[2]) and np.array_equal(bo.plane(0).cpu().numpy(), ho[0]), tag + ": an input was modified"
            # the harness entries: generators, io <-> hd
            g = mk(); vg = g.view()
            _lib.check(lib.djb_gen_directions(gpu_ctx._h, C.c_int64(n), C.c_uint32(3), C.c_uint64(5), C.byref(vg)))
            gu = mk(1)
            _lib.check(lib.djb_gen_uniforms(gpu_ctx._h, C.c_int64(n), C.c_uint32(3), C.c_uint64(5), C.c_void_p(gu.ptr())))
            bi, bo, oh, od = mk(fill=hi), mk(fill=ho), mk(), mk()
            vi, vo, vh, vd = bi.view(), bo.view(), oh.view(), od.view()
            _lib.check(lib.djb_io_to_hd_batch(gpu_ctx._h, C.c_int64(n), C.byref(vi), C.byref(vo), C.byref(vh), C.byref(vd), C.c_int(_lib.MEM_DEVICE)))
            torch.cuda.synchronize()
            g.check(f"gen_directions n={n}"); gu.check(f"gen_uniforms n={n}"); oh.check(f"io_to_hd h n={n}"); od.check(f"io_to_hd d n={n}")
    finally:
        djb.set_contract_1e5(gpu_ctx, False)


def test_lean_batches_stay_inside_their_outputs(gpu_ctx):
    import torch
    lib = _lib.load()
    dev = f"cuda:{gpu_ctx.device}"
    b = djb.beckmann(djb.fresnel.schlick((0.9, 0.8, 0.7)), ctx=gpu_ctx)
    base = djb.microfacet.params.elliptic(0.3, 0.2, 0.4)
    rng = np.random.default_rng(3)
    for n in SIZES:
        hi, ho = synth.directions(n, 21), synth.directions(n, 22)
        lean = np.concatenate([rng.uniform(24.8, 25.2, (n, 2)), rng.uniform(625.0, 625.3, (n, 2)), rng.uniform(624.9, 625.1, (n, 1))], 1).astype(np.float32)
        bl = Buf(torch, dev, 5 * n, 1, fill=[lean.reshape(-1)])
        bi, bo = Buf(torch, dev, n, fill=hi), Buf(torch, dev, n, fill=ho)
        bu1, bu2 = Buf(torch, dev, n, 1, fill=[synth.uniforms(n, 23)]), Buf(torch, dev, n, 1, fill=[synth.uniforms(n, 24)])
        vi, vo = bi.view(), bo.view()
        out, pdf, pp = Buf(torch, dev, n), Buf(torch, dev, n, 1), Buf(torch, dev, 5 * n, 1)
        vout = out.view()
        _lib.check(lib.djb_eval_lean_batch(gpu_ctx._h, b._h, C.c_int64(n), C.byref(vi), C.byref(vo), C.byref(base._p), C.c_float(0.8), C.c_int(2), C.c_void_p(bl.ptr()),
                                           C.c_int(6), C.byref(vout), C.c_void_p(pdf.ptr()), C.c_void_p(pp.ptr()), C.c_int(_lib.MEM_DEVICE)))
        torch.cuda.synchronize(); out.check(f"eval_lean n={n}"); pdf.check(f"eval_lean pdf n={n}"); pp.check(f"eval_lean params n={n}")
        ow, oi, pdf, pp = Buf(torch, dev, n), Buf(torch, dev, n), Buf(torch, dev, n, 1), Buf(torch, dev, 5 * n, 1)
        vow, voi = ow.view(), oi.view()
        _lib.check(lib.djb_sample_lean_batch(gpu_ctx._h, b._h, C.c_int64(n), C.c_void_p(bu1.ptr()), C.c_void_p(bu2.ptr()), C.byref(vo), C.byref(base._p), C.c_float(0.8),
                                             C.c_int(2), C.c_void_p(bl.ptr()), C.byref(vow), C.byref(voi), C.c_void_p(pdf.ptr()), C.c_void_p(pp.ptr()), C.c_int(_lib.MEM_DEVICE)))
        torch.cuda.synchronize(); ow.check(f"sample_lean w n={n}"); oi.check(f"sample_lean i n={n}"); pdf.check(f"sample_lean pdf n={n}"); pp.check(f"sample_lean params n={n}")
        bl.check(f"lean input n={n}", written=False)
