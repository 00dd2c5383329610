"""Host-side mirror of the reference's operator interface (namespace ``djb``, dj_brdf.h:41-537)
for the hot path, on top of the C ABI in include/djb_hip.h.

Same names and argument meaning as the reference: ``brdf.eval(i, o, user_param)``,
``evalp``, ``pdf``, ``sample(u1, u2, o, user_param)``, ``evalp_is``, ``brdf.io_to_hd`` /
``hd_to_io``; ``merl(filename)``, ``utia(filename)``, ``beckmann(fresnel, shadow)``,
``ggx(fresnel, shadow)``, ``tabular(brdf, res, shadow)`` with ``fit_beckmann_parameters`` /
``fit_ggx_parameters`` / ``get_p22v`` ...; ``microfacet.params.{standard,isotropic,elliptic,
pdfparams}``; ``fresnel.{ideal,unpolarized,schlick,sgd,spline}``; errors raise ``exc``.
The one difference is cardinality: every direction argument is a BATCH.

Array conventions
  * numpy float32 ``[n, 3]`` (array of djb::vec3) or ``[3, n]`` (SoA): host memory, staged
    through HBM by the library; results come back as numpy in the same layout.
  * torch CUDA float32 tensors of the same shapes: device memory, zero-copy, asynchronous on
    torch's current stream; results are torch tensors on the same device.  ``[3, n]`` is the
    coalesced fast path.
All arithmetic happens in the HIP kernels; nothing here evaluates a BRDF on the CPU.
"""
from __future__ import annotations

import threading
import ctypes as C
from typing import Optional, Sequence

import numpy as np

from . import _lib
from ._lib import exc  # noqa: F401  (re-export: djb.exc)

try:  # plumbing only: device memory + streams
    import torch
except Exception:  # pragma: no cover
    torch = None


# --------------------------------------------------------------------------- context
class Context:
    """One GPU + one HIP stream (``djb_ctx``).  With torch present the ctx runs on torch's
    current stream of that device, so torch events/synchronisation see the kernels."""

    def __init__(self, device=0, stream: Optional[int] = None):
        lib = _lib.load()
        # device "cpu" (DJB_DEVICE_CPU = -1): the product's HOST execution path -- the kernels' per-unit code compiled for
        # the CPU with the host libm; explicit only, a GPU context never falls back to it for batches
        self.is_cpu = device in ("cpu", -1)
        if self.is_cpu:
            device, stream = -1, None
        self.device = device
        self._handle = C.c_void_p()
        if self.is_cpu:
            self._follow_torch, self._stream = False, None
            _lib.check(lib.djb_ctx_create(C.c_int(-1), C.byref(self._handle)))
            return
        # stream=None with torch present: the ctx FOLLOWS torch's current stream of the device -- re-read on every
        # call (the `_h` property), so work issued under `with torch.cuda.stream(s):` runs on s, where torch
        # allocated the outputs and where its consumers wait.  An explicit stream pins the ctx to it.
        self._follow_torch = stream is None and torch is not None and torch.cuda.is_available()
        if self._follow_torch:
            stream = torch.cuda.current_stream(device).cuda_stream   # 0 == the default (null) stream
        self._stream = stream
        if stream is None:
            _lib.check(lib.djb_ctx_create(C.c_int(device), C.byref(self._handle)))
        else:
            _lib.check(lib.djb_ctx_create_on_stream(C.c_int(device), C.c_void_p(stream), C.byref(self._handle)))

    @property
    def _h(self):
        if self._follow_torch and self._handle:
            cur = torch.cuda.current_stream(self.device).cuda_stream
            if cur != self._stream:
                _lib.check(_lib.load().djb_ctx_set_stream(self._handle, C.c_void_p(cur)))
                self._stream = cur
        return self._handle

    def synchronize(self):
        _lib.check(_lib.load().djb_ctx_synchronize(self._h))

    def timer_start(self):
        _lib.check(_lib.load().djb_timer_start(self._h))

    def timer_stop_ms(self) -> float:
        ms = C.c_float()
        _lib.check(_lib.load().djb_timer_stop_ms(self._h, C.byref(ms)))
        return ms.value

    def close(self):
        if self._handle:
            _lib.load().djb_ctx_destroy(self._handle)
            self._handle = C.c_void_p()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


_default_cpu_ctx = {}
_default_gpu_ctx = threading.local()


def cpu_context() -> Context:
    """The process-wide CPU context (Context("cpu"))."""
    return default_context("cpu")


def default_context(device=0) -> Context:
    """The default context of `device`: process-wide for "cpu", PER THREAD for a GPU.  A GPU context created without an
    explicit stream follows torch's current stream, which is a per-thread notion: `set stream, then launch` on a context
    shared by two threads that work under different torch streams could interleave (thread A's kernels on thread B's
    stream).  Per-thread defaults keep threads that each BUILD their own objects apart.  What they do not do: a BRDF object
    remembers the context it was built with (`obj.ctx`) and its methods launch on that one, so an object created on
    thread A and called from thread B still drives A's default context and follows whatever torch stream the CALLING
    thread has current.  Sharing objects across threads that use different torch streams therefore needs either a pinned
    stream (Context(device, stream=...), passed as ctx= at construction) or rebinding (`obj.ctx = default_context(dev)` on
    the calling thread -- handles may be used with any context of their device, tests/test_gpu_scalar_path.py)."""
    if device in ("cpu", -1):
        if "cpu" not in _default_cpu_ctx:
            _default_cpu_ctx["cpu"] = Context("cpu")
        return _default_cpu_ctx["cpu"]
    d = _default_gpu_ctx.__dict__.setdefault("ctx", {})
    if device not in d:
        d[device] = Context(device)
    return d[device]


_MAP_OBSERVER = C.CFUNCTYPE(None, C.c_char_p, C.c_void_p)


def set_file_map_observer(fn):
    """djb_set_file_map_observer: ``fn(path)`` is called by the MERL file pipeline after a file has been size-checked and mapped,
    before its entries are gathered (None removes it).  Returns the ctypes callback: the caller keeps it alive."""
    lib = _lib.load()
    if fn is None:
        _lib.check(lib.djb_set_file_map_observer(None, None))
        return None
    cb = _MAP_OBSERVER(lambda path, _user: fn(path.decode()))
    _lib.check(lib.djb_set_file_map_observer(cb, None))
    return cb


def device_count() -> int:
    n = C.c_int()
    st = _lib.load().djb_device_count(C.byref(n))
    return n.value if st == 0 else 0


# --------------------------------------------------------------------------- array plumbing
class _Vec:
    """A vec3 batch resolved to (view, memory space, n, how to build a like-shaped output)."""

    def __init__(self, a, n_hint=None):
        self.is_torch = torch is not None and isinstance(a, torch.Tensor)
        if self.is_torch and not a.is_cuda:
            a = a.numpy()
            self.is_torch = False
        if self.is_torch:
            if a.dtype != torch.float32:
                a = a.float()
            a = a.contiguous()
            shape = tuple(a.shape)
            base, itemsz = a.data_ptr(), 4
            self.mem = _lib.MEM_DEVICE
            self.device = a.device
        else:
            a = np.ascontiguousarray(a, dtype=np.float32)
            shape = a.shape
            base, itemsz = a.ctypes.data, 4
            self.mem = _lib.MEM_HOST
            self.device = None
        if len(shape) != 2 or (shape[1] != 3 and shape[0] != 3):
            raise exc(1, f"djb_error: expected a [n,3] or [3,n] float32 array, got {shape}")
        self.aos = shape[1] == 3
        self.n = shape[0] if self.aos else shape[1]
        self.keep = a
        v = _lib.Vec3View()
        if self.aos:
            v.x, v.y, v.z, v.stride = base, base + itemsz, base + 2 * itemsz, 3
        else:
            v.x, v.y, v.z, v.stride = base, base + itemsz * self.n, base + 2 * itemsz * self.n, 1
        self.view = v

    def like(self):
        """An uninitialised output batch with the same framework / layout / device."""
        shape = (self.n, 3) if self.aos else (3, self.n)
        if self.is_torch:
            return _Vec(torch.empty(shape, dtype=torch.float32, device=self.device))
        return _Vec(np.empty(shape, dtype=np.float32))

    def scalars(self, dtype=np.float32):
        if self.is_torch:
            tdt = torch.float32 if dtype == np.float32 else torch.int32
            t = torch.empty((self.n,), dtype=tdt, device=self.device)
            return t, t.data_ptr()
        a = np.empty((self.n,), dtype=dtype)
        return a, a.ctypes.data


def _scalar_in(u, like: _Vec):
    if like.is_torch:
        if not (torch is not None and isinstance(u, torch.Tensor) and u.is_cuda):
            u = torch.as_tensor(np.asarray(u, dtype=np.float32), device=like.device)
        u = u.float().contiguous()
        return u, u.data_ptr()
    u = np.ascontiguousarray(u, dtype=np.float32)
    return u, u.ctypes.data


# --------------------------------------------------------------------------- fresnel (dj_brdf.h:149-207)
def vec3_from_angles(theta, phi) -> np.ndarray:
    """``djb::vec3(theta, phi)`` (dj_brdf.h:67, 589-595): [n,3] unit vectors, the reference's float/double order
    (s = float(sin(theta)); x = float(s * cos(phi)); y = float(s * sin(phi)); z = float(cos(theta)))."""
    t = np.atleast_1d(np.asarray(theta, dtype=np.float32)).astype(np.float64)
    p = np.atleast_1d(np.asarray(phi, dtype=np.float32)).astype(np.float64)
    s = np.sin(t).astype(np.float32).astype(np.float64)
    return np.stack([(s * np.cos(p)).astype(np.float32), (s * np.sin(p)).astype(np.float32), np.cos(t).astype(np.float32)], 1)


class fresnel:
    @staticmethod
    def ior_to_f0(ior):
        """fresnel::ior_to_f0 (dj_brdf.h:1255-1270): ((ior - 1) / (ior + 1))^2 with the reference's
        float/double evaluation order; scalar or array."""
        x = np.asarray(ior, dtype=np.float32)
        tmp = ((x.astype(np.float64) - 1.0) / (x.astype(np.float64) + 1.0)).astype(np.float32)
        r = tmp * tmp
        return r if r.ndim else np.float32(r)

    @staticmethod
    def f0_to_ior(f0):
        """fresnel::f0_to_ior (dj_brdf.h:1272-1290)."""
        x = np.asarray(f0, dtype=np.float32)
        s = np.sqrt(x.astype(np.float64)).astype(np.float32).astype(np.float64)
        with np.errstate(divide="ignore", invalid="ignore"):
            r = np.where(x.astype(np.float64) == 1.0, 1.0, (1.0 + s) / (1.0 - s)).astype(np.float32)
        return r if r.ndim else np.float32(r)

    class impl:
        kind = 0

        def _desc(self):
            d = _lib.FresnelDesc()
            d.kind = self.kind
            return d, None

        def eval(self, cos_theta_d, ctx=None):
            """fresnel::impl::eval (dj_brdf.h:160) on a batch of cosines -> [n,3]; evaluated through a
            temporary microfacet object (inside a BRDF the term is fused into the eval kernels)."""
            return ggx(self, True, ctx=ctx).fresnel(cos_theta_d)

    class ideal(impl):
        kind = 0

    class unpolarized(impl):
        kind = 1

        def __init__(self, ior: Sequence[float]):
            self.ior = tuple(float(x) for x in ior)

        def _desc(self):
            d, _ = super()._desc()
            d.a[:] = self.ior
            return d, None

    class schlick(impl):
        kind = 2

        def __init__(self, f0: Sequence[float]):
            self.f0 = tuple(float(x) for x in f0)

        def _desc(self):
            d, _ = super()._desc()
            d.a[:] = self.f0
            return d, None

    class sgd(impl):
        kind = 3

        def __init__(self, f0: Sequence[float], f1: Sequence[float]):
            self.f0, self.f1 = tuple(map(float, f0)), tuple(map(float, f1))

        def _desc(self):
            d, _ = super()._desc()
            d.a[:] = self.f0
            d.b[:] = self.f1
            return d, None

    class spline(impl):
        kind = 4

        def __init__(self, points):
            self.points = np.ascontiguousarray(points, dtype=np.float32).reshape(-1, 3)

        def get_points(self):
            return self.points

        def _desc(self):
            d, _ = super()._desc()
            d.points = self.points.ctypes.data
            d.npoints = self.points.shape[0]
            return d, self.points

    @staticmethod
    def ior_to_f0(ior):  # dj_brdf.h:1255-1262
        ior = np.asarray(ior, dtype=np.float32)
        tmp = ((ior.astype(np.float64) - 1.0) / (ior.astype(np.float64) + 1.0)).astype(np.float32)
        return tmp * tmp

    @staticmethod
    def f0_to_ior(f0):  # dj_brdf.h:1272-1282
        f0 = np.asarray(f0, dtype=np.float32)
        s = np.sqrt(f0.astype(np.float64)).astype(np.float32).astype(np.float64)
        with np.errstate(divide="ignore", invalid="ignore"):
            out = ((1.0 + s) / (1.0 - s)).astype(np.float32)
        return np.where(f0 == 1.0, np.float32(1.0), out)


# --------------------------------------------------------------------------- brdf base (dj_brdf.h:74-109)
class brdf:
    def __init__(self, ctx: Optional[Context]):
        self.ctx = ctx or default_context()
        self._h = C.c_void_p()

    # ---- operator surface
    def _eval(self, fn_name, i, o, user_param, want_pdf=False, want_fr=True, want_cos=0):
        lib = _lib.load()
        vi, vo = _Vec(i), _Vec(o)
        if vi.n != vo.n or vi.mem != vo.mem:
            raise exc(1, "djb_error: i and o must have the same length and memory space")
        p = _params_ptr(user_param)
        out = vi.like() if want_fr else None
        pdf = pdf_ptr = None
        if want_pdf:
            pdf, pdf_ptr = vi.scalars()
        n = C.c_int64(vi.n)
        if fn_name == "djb_pdf_batch":
            st = lib.djb_pdf_batch(self.ctx._h, self._h, n, C.byref(vi.view), C.byref(vo.view), p,
                                   C.c_void_p(pdf_ptr), C.c_int(vi.mem))
        elif fn_name == "djb_eval_pdf_batch":
            st = lib.djb_eval_pdf_batch(self.ctx._h, self._h, n, C.byref(vi.view), C.byref(vo.view), p,
                                        C.c_int(want_cos), C.byref(out.view), C.c_void_p(pdf_ptr),
                                        C.c_int(vi.mem))
        else:
            st = getattr(lib, fn_name)(self.ctx._h, self._h, n, C.byref(vi.view), C.byref(vo.view), p,
                                       C.byref(out.view), C.c_int(vi.mem))
        _lib.check(st)
        if want_fr and want_pdf:
            return out.keep, pdf
        return out.keep if want_fr else pdf

    def eval(self, i, o, user_param=None):
        """f_r for every pair (dj_brdf.h:77-78)."""
        return self._eval("djb_eval_batch", i, o, user_param)

    def evalp(self, i, o, user_param=None):
        """f_r * cos(theta_i) (dj_brdf.h:82-83)."""
        return self._eval("djb_evalp_batch", i, o, user_param)

    def pdf(self, i, o, user_param=None):
        """pdf of sampling i given o (dj_brdf.h:96-97)."""
        return self._eval("djb_pdf_batch", i, o, user_param, want_pdf=True, want_fr=False)

    def eval_pdf(self, i, o, user_param=None, cos=False):
        """eval (or evalp) and pdf in one pass over HBM."""
        return self._eval("djb_eval_pdf_batch", i, o, user_param, want_pdf=True, want_cos=int(cos))

    def sample(self, u1, u2, o, user_param=None):
        """importance-sample i from two uniform numbers per element (dj_brdf.h:92-94)."""
        lib = _lib.load()
        vo = _Vec(o)
        k1, p1 = _scalar_in(u1, vo)
        k2, p2 = _scalar_in(u2, vo)
        out = vo.like()
        _lib.check(lib.djb_sample_batch(self.ctx._h, self._h, C.c_int64(vo.n), C.c_void_p(p1), C.c_void_p(p2),
                                        C.byref(vo.view), _params_ptr(user_param), C.byref(out.view),
                                        C.c_int(vo.mem)))
        del k1, k2
        return out.keep

    def sample_rng(self, seed_u1: int, seed_u2: int, o, user_param=None, start: int = 0):
        """sample() with the uniforms drawn on chip (device arrays only)."""
        lib = _lib.load()
        vo = _Vec(o)
        if vo.mem != _lib.MEM_DEVICE and not self.ctx.is_cpu:
            raise exc(1, "djb_error: sample_rng needs device arrays")
        out = vo.like()
        _lib.check(lib.djb_sample_rng_batch(self.ctx._h, self._h, C.c_int64(vo.n), C.c_uint32(seed_u1),
                                            C.c_uint32(seed_u2), C.c_uint64(start), C.byref(vo.view),
                                            _params_ptr(user_param), C.byref(out.view)))
        return out.keep

    def evalp_is(self, u1, u2, o, user_param=None):
        """(f_r cos / pdf, i, pdf) (dj_brdf.h:87-90)."""
        lib = _lib.load()
        vo = _Vec(o)
        k1, p1 = _scalar_in(u1, vo)
        k2, p2 = _scalar_in(u2, vo)
        w, i = vo.like(), vo.like()
        pdf, pdf_ptr = vo.scalars()
        _lib.check(lib.djb_evalp_is_batch(self.ctx._h, self._h, C.c_int64(vo.n), C.c_void_p(p1), C.c_void_p(p2),
                                          C.byref(vo.view), _params_ptr(user_param), C.byref(w.view),
                                          C.byref(i.view), C.c_void_p(pdf_ptr), C.c_int(vo.mem)))
        del k1, k2
        return w.keep, i.keep, pdf

    # ---- static utilities (dj_brdf.h:99-100)
    @staticmethod
    def io_to_hd(i, o, ctx: Optional[Context] = None):
        return _hd("djb_io_to_hd_batch", i, o, ctx)

    @staticmethod
    def hd_to_io(h, d, ctx: Optional[Context] = None):
        return _hd("djb_hd_to_io_batch", h, d, ctx)

    def close(self):
        if self._h:
            _lib.load().djb_brdf_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


def _hd(fn, a, b, ctx):
    lib = _lib.load()
    ctx = ctx or default_context()
    va, vb = _Vec(a), _Vec(b)
    oc, od = va.like(), va.like()
    _lib.check(getattr(lib, fn)(ctx._h, C.c_int64(va.n), C.byref(va.view), C.byref(vb.view),
                                C.byref(oc.view), C.byref(od.view), C.c_int(va.mem)))
    return oc.keep, od.keep


def merl_index(i, o, ctx: Optional[Context] = None):
    """The table index merl::eval composes for each pair (dj_brdf.h:997-1002)."""
    lib = _lib.load()
    ctx = ctx or default_context()
    vi, vo = _Vec(i), _Vec(o)
    idx, ptr = vi.scalars(np.int32)
    _lib.check(lib.djb_merl_index_batch(ctx._h, C.c_int64(vi.n), C.byref(vi.view), C.byref(vo.view),
                                        C.c_void_p(ptr), C.c_int(vi.mem)))
    return idx


def merl_bin_keys(i, o, ctx: Optional[Context] = None):
    """21-bit MERL bin keys from the look-up's tier-1 arithmetic (djb_merl_bin_keys_batch): == merl_index where tier 1 is certain
    (99.6 % of random pairs), the neighbouring bin otherwise.  For ordering a batch before merl.eval; int32 array / tensor."""
    lib = _lib.load()
    ctx = ctx or default_context()
    vi, vo = _Vec(i), _Vec(o)
    keys, ptr = vi.scalars(np.int32)
    _lib.check(lib.djb_merl_bin_keys_batch(ctx._h, C.c_int64(vi.n), C.byref(vi.view), C.byref(vo.view),
                                           C.c_void_p(ptr), C.c_int(vi.mem)))
    return keys


# --------------------------------------------------------------------------- microfacet (dj_brdf.h:210-298)
class microfacet(brdf):
    class params:
        """djb::microfacet::params (dj_brdf.h:213-243); build with the factories."""

        def __init__(self, kind: int, values: Sequence[float]):
            self._p = _lib.Params()
            self._p.kind = kind
            for k, v in enumerate(values):
                self._p.v[k] = v

        @staticmethod
        def standard():
            return microfacet.params(0, ())

        @staticmethod
        def isotropic(a):
            return microfacet.params(1, (a, a, 0.0))

        @staticmethod
        def elliptic(a1, a2, phi_a=0.0):
            return microfacet.params(1, (a1, a2, phi_a))

        @staticmethod
        def pdfparams(ax, ay, rho=0.0, tx_n=0.0, ty_n=0.0):
            return microfacet.params(2, (ax, ay, rho, tx_n, ty_n))

        def _resolved(self):
            r = _lib.ParamsResolved()
            _lib.check(_lib.load().djb_params_resolve(C.byref(self._p), C.byref(r)))
            return r

        def get_ellipse(self):
            r = self._resolved()
            return r.a1, r.a2, r.phi_a

        def get_pdfparams(self):
            r = self._resolved()
            return r.ax, r.ay, r.rho, r.tx_n, r.ty_n

        def get_location(self):
            r = self._resolved()
            return tuple(r.n)

    def __init__(self, kind_fn: str, fresnel_impl, shadow: bool, ctx):
        super().__init__(ctx)
        fresnel_impl = fresnel_impl or fresnel.ideal()
        d, keep = fresnel_impl._desc()
        _lib.check(getattr(_lib.load(), kind_fn)(self.ctx._h, C.byref(d), C.c_int(int(shadow)), C.byref(self._h)))
        self._fresnel = fresnel_impl
        del keep

    # ---- eval / sampling queries (dj_brdf.h:258-276, 307-314), batched
    def _query(self, which: int, a, b=None, c=None, params=None, width=1):
        lib = _lib.load()
        va = _Vec(a)
        vb = _Vec(b) if b is not None else None
        vc = _Vec(c) if c is not None else None
        out = va.like()
        _lib.check(lib.djb_query_batch(self.ctx._h, self._h, C.c_int(which), C.c_int64(va.n), C.byref(va.view),
                                       C.byref(vb.view) if vb else None, C.byref(vc.view) if vc else None,
                                       _params_ptr(params), C.byref(out.view), C.c_int(va.mem)))
        r = out.keep
        if width == 3:
            return r
        return r[:, 0] if out.aos else r[0]

    @staticmethod
    def _cols(*cols):
        """pack up to three scalar arrays into a [n,3] batch (missing columns are zero)."""
        first = cols[0]
        if torch is not None and isinstance(first, torch.Tensor) and first.is_cuda:
            z = torch.zeros_like(first, dtype=torch.float32)
            return torch.stack([(cols[k].float() if k < len(cols) else z) for k in range(3)], dim=0)
        first = np.asarray(first, dtype=np.float32)
        z = np.zeros_like(first)
        return np.stack([(np.asarray(cols[k], np.float32) if k < len(cols) else z) for k in range(3)], axis=1)

    def fresnel(self, cos_theta_d):
        return self._query(7, self._cols(cos_theta_d), width=3)

    def ndf(self, h, params=None):
        return self._query(0, h, params=params)

    def gaf(self, h, i, o, params=None):
        return self._query(1, h, i, o, params)

    def g1(self, h, k, params=None):
        return self._query(2, h, k, params=params)

    def sigma(self, k, params=None):
        return self._query(3, k, params=params)

    def p22(self, x, y, params=None):
        return self._query(4, self._cols(x, y), params=params)

    def vp22(self, x, y, k, params=None):
        return self._query(5, self._cols(x, y), k, params=params)

    def vndf(self, h, k, params=None):
        return self._query(6, h, k, params=params)

    # radial (dj_brdf.h:307-314) -- beckmann / ggx / tabular
    def p22_radial(self, r_sqr):
        return self._query(16, self._cols(r_sqr))

    def sigma_std_radial(self, cos_theta_k):
        return self._query(17, self._cols(cos_theta_k))

    def cdf_radial(self, r):
        return self._query(18, self._cols(r))

    def qf_radial(self, u):
        return self._query(19, self._cols(u))

    def qf2_radial(self, u, cos_theta_k, sin_theta_k):
        return self._query(20, self._cols(u, cos_theta_k, sin_theta_k))

    def qf3_radial(self, u, qf2):
        return self._query(21, self._cols(u, qf2))

    def qf1(self, u):
        return self._query(22, self._cols(u))

    def qf2(self, u, k):
        """microfacet::qf2 base stub (dj_brdf.h:1783-1786)"""
        raise exc(5, "djb_error: Not Implemented")

    def qf3(self, u, k, qf2):
        """microfacet::qf3 base stub (dj_brdf.h:1788-1791)"""
        raise exc(5, "djb_error: Not Implemented")

    def get_shadow(self) -> int:
        return _lib.load().djb_brdf_get_shadow(self._h)

    def set_shadow(self, shadow: bool) -> None:
        """microfacet::set_shadow, dj_brdf.h:278"""
        _lib.check(_lib.load().djb_brdf_set_shadow(self._h, C.c_int(1 if shadow else 0)))

    def set_fresnel(self, fresnel_impl) -> None:
        """microfacet::set_fresnel, dj_brdf.h:279 (e.g. ``tab.set_fresnel(fresnel.ideal())``)"""
        d, keep = fresnel_impl._desc()
        _lib.check(_lib.load().djb_brdf_set_fresnel(self._h, C.byref(d)))
        self._fresnel = fresnel_impl
        self._fresnel_replaced = True

    def get_fresnel(self):
        return self._fresnel


def _params_ptr(user_param):
    if user_param is None:
        return None
    if not isinstance(user_param, (microfacet.params, lambert.params)):
        raise exc(1, "djb_error: user_param must be a microfacet.params, a lambert.params or None")
    return C.byref(user_param._p)


def _eval_records(self, fn, i, o, records, want, extra_args, want_pp=False):
    """shared driver of eval_pp / eval_lean: per-pair 5-float records next to the directions."""
    lib = _lib.load()
    vi, vo = _Vec(i), _Vec(o)
    if vi.is_torch:
        rec = records.float().contiguous() if isinstance(records, torch.Tensor) else \
            torch.as_tensor(np.asarray(records, np.float32), device=vi.device)
        rec_ptr = rec.data_ptr()
    else:
        rec = np.ascontiguousarray(records, dtype=np.float32)
        rec_ptr = rec.ctypes.data
    assert tuple(rec.shape) == (vi.n, 5), "records must be [n, 5]"
    code = {"eval": 1, "evalp": 2, "pdf": 4, "eval+pdf": 5, "evalp+pdf": 6}[want]
    out = vi.like() if code & 3 else None
    pdf = pdf_ptr = None
    if code & 4:
        pdf, pdf_ptr = vi.scalars()
    pp = pp_ptr = None
    if want_pp:
        pp = torch.empty((vi.n, 5), dtype=torch.float32, device=vi.device) if vi.is_torch else np.empty((vi.n, 5), np.float32)
        pp_ptr = pp.data_ptr() if vi.is_torch else pp.ctypes.data
    args = [self.ctx._h, self._h, C.c_int64(vi.n), C.byref(vi.view), C.byref(vo.view)] + extra_args(rec_ptr) + \
           [C.c_int(code), C.byref(out.view) if out else None, C.c_void_p(pdf_ptr)]
    if fn == "djb_eval_lean_batch":
        args.append(C.c_void_p(pp_ptr))
    args.append(C.c_int(vi.mem))
    _lib.check(getattr(lib, fn)(*args))
    res = [x for x in (out.keep if out else None, pdf) if x is not None]
    if want_pp:
        res.append(pp)
    return res[0] if len(res) == 1 else tuple(res)


def _eval_pp(self, i, o, pdfparams, want="evalp"):
    """per-pair microfacet::params::pdfparams records [n,5] = (ax, ay, rho, tx_n, ty_n)."""
    return _eval_records(self, "djb_eval_pp_batch", i, o, pdfparams, want, lambda p: [C.c_void_p(p)])


LEAN_NAIVE_MIP, LEAN_BIASED = 1, 2      # include/djb_hip.h DJB_LEAN_*


def _eval_lean(self, i, o, base, scale, lean, want="evalp", return_params=False, filtering=True, biased=False):
    """dj_beckmannconductor's per-hit path, batched (mitsuba/dj_beckmannconductor.cpp:296-314):
    params_k = lrep_to_params(lrep(lean_k) * scale + params_to_lrep(base)), lean [n,5] = LEAN/LEADR texel moments
    (E1..E5), scale = the plugin's dmapscale.  filtering=False: the plugin's leanFiltering=false branch,
    lrep(E1, E2, E1*E1, E2*E2, E1*E2).  biased=True: records are raw texels (E1, E2 carry +25, E5 +625)."""
    flags = (0 if filtering else LEAN_NAIVE_MIP) | (LEAN_BIASED if biased else 0)
    return _eval_records(self, "djb_eval_lean_batch", i, o, lean, want,
                         lambda p: [_params_ptr(base), C.c_float(scale), C.c_int(flags), C.c_void_p(p)], want_pp=return_params)


def _sample_records(self, fn, u1, u2, o, records, evalp_is, extra_args, want_pp=False):
    """shared driver of sample_pp / sample_lean: sample() or evalp_is() with per-pair 5-float records."""
    lib = _lib.load()
    vo = _Vec(o)
    k1, p1 = _scalar_in(u1, vo)
    k2, p2 = _scalar_in(u2, vo)
    if vo.is_torch:
        rec = records.float().contiguous() if isinstance(records, torch.Tensor) else \
            torch.as_tensor(np.asarray(records, np.float32), device=vo.device)
        rec_ptr = rec.data_ptr()
    else:
        rec = np.ascontiguousarray(records, dtype=np.float32)
        rec_ptr = rec.ctypes.data
    assert tuple(rec.shape) == (vo.n, 5), "records must be [n, 5]"
    i = vo.like()
    w = vo.like() if evalp_is else None
    pdf, pdf_ptr = vo.scalars() if evalp_is else (None, None)
    pp = pp_ptr = None
    if want_pp:
        pp = torch.empty((vo.n, 5), dtype=torch.float32, device=vo.device) if vo.is_torch else np.empty((vo.n, 5), np.float32)
        pp_ptr = pp.data_ptr() if vo.is_torch else pp.ctypes.data
    args = [self.ctx._h, self._h, C.c_int64(vo.n), C.c_void_p(p1), C.c_void_p(p2), C.byref(vo.view)] + extra_args(rec_ptr) + \
           [C.byref(w.view) if w else None, C.byref(i.view), C.c_void_p(pdf_ptr)]
    if fn == "djb_sample_lean_batch":
        args.append(C.c_void_p(pp_ptr))
    args.append(C.c_int(vo.mem))
    _lib.check(getattr(lib, fn)(*args))
    del k1, k2
    res = (w.keep, i.keep, pdf) if evalp_is else (i.keep,)
    if want_pp:
        res = res + (pp,)
    return res[0] if len(res) == 1 else res


def _sample_pp(self, u1, u2, o, pdfparams, evalp_is=False):
    """sample() -- or with evalp_is=True (weight, i, pdf) -- with per-pair pdfparams records [n,5]."""
    return _sample_records(self, "djb_sample_pp_batch", u1, u2, o, pdfparams, evalp_is, lambda p: [C.c_void_p(p)])


def _sample_lean(self, u1, u2, o, base, scale, lean, evalp_is=True, return_params=False, filtering=True, biased=False):
    """dj_beckmann_conductor::sample, batched (mitsuba/dj_beckmannconductor.cpp:373-413): per-pair params as eval_lean,
    then evalp_is (or sample) with them."""
    flags = (0 if filtering else LEAN_NAIVE_MIP) | (LEAN_BIASED if biased else 0)
    return _sample_records(self, "djb_sample_lean_batch", u1, u2, o, lean, evalp_is,
                           lambda p: [_params_ptr(base), C.c_float(scale), C.c_int(flags), C.c_void_p(p)], want_pp=return_params)


microfacet.eval_pp = _eval_pp
microfacet.eval_lean = _eval_lean
microfacet.sample_pp = _sample_pp
microfacet.sample_lean = _sample_lean


class beckmann(microfacet):
    class lrep:
        """beckmann::lrep: linear representation by slope moments E1..E5 (dj_brdf.h:330-356)."""

        def __init__(self, E1=0.0, E2=0.0, E3=1.0, E4=1.0, E5=0.0):
            self.E = (C.c_float * 5)(E1, E2, E3, E4, E5)

        def _op(self, op, other=None, x=0.0, y=0.0):
            out = beckmann.lrep()
            _lib.check(_lib.load().djb_lrep_op(C.c_int(op), self.E, other.E if other is not None else None,
                                              C.c_float(x), C.c_float(y), out.E))
            return out

        def __add__(self, r):
            return self._op(0, r)

        def __mul__(self, sc):
            return self._op(1, None, sc)

        def __iadd__(self, r):
            self.E = self._op(2, r).E
            return self

        def __imul__(self, sc):
            self.E = self._op(3, None, sc).E
            return self

        def shear(self, x, y):
            self.E = self._op(4, None, x, y).E

        def scale(self, x, y):
            self.E = self._op(5, None, x, y).E

        def moments(self):
            return tuple(self.E)

    @staticmethod
    def params_to_lrep(params):
        l = beckmann.lrep()
        _lib.check(_lib.load().djb_params_to_lrep(_params_ptr(params), l.E))
        return l

    @staticmethod
    def lrep_to_params(l):
        p = microfacet.params(0, ())
        _lib.check(_lib.load().djb_lrep_to_params(l.E, C.byref(p._p)))
        return p

    def __init__(self, fresnel=None, shadow=True, ctx=None):
        super().__init__("djb_brdf_create_beckmann", fresnel, shadow, ctx)

    def supports_smith_vndf_sampling(self):
        return True


class ggx(microfacet):
    def __init__(self, fresnel=None, shadow=True, ctx=None):
        super().__init__("djb_brdf_create_ggx", fresnel, shadow, ctx)

    def supports_smith_vndf_sampling(self):
        return True


class lambert(brdf):
    class params:
        """lambert::params(reflectance) (dj_brdf.h:114-119); pass as ``user_param``."""

        def __init__(self, reflectance=(1.0, 1.0, 1.0)):
            self.m_reflectance = tuple(float(x) for x in reflectance)
            self._p = _lib.Params()
            self._p.kind = 3
            for k, v in enumerate(self.m_reflectance):
                self._p.v[k] = v

    def __init__(self, ctx=None):
        super().__init__(ctx)
        _lib.check(_lib.load().djb_brdf_create_lambert(self.ctx._h, C.byref(self._h)))


class _model_queries:
    """ndf / gaf / g1 / fresnel members of the sgd and abc classes (dj_brdf.h:505-509, 530-533)."""

    def _mq(self, which, a, b=None, c=None, width=3):
        return microfacet._query(self, which, a, b, c, None, width)

    def ndf(self, h):
        return self._mq(48, h)

    def fresnel(self, cos_theta_d):
        return self._mq(7, microfacet._cols(cos_theta_d))


class sgd(brdf, _model_queries):
    """djb::sgd(name): Shifted Gamma Distribution BRDF with the published per-material parameters
    (dj_brdf.h:481-511, 3436-3500).  Unknown names raise exc ("No SGD parameters for ...")."""

    def __init__(self, name: str, ctx=None):
        super().__init__(ctx)
        _lib.check(_lib.load().djb_brdf_create_sgd(self.ctx._h, name.encode(), C.byref(self._h)))

    @classmethod
    def from_params(cls, params33, ctx=None):
        self = cls.__new__(cls)
        brdf.__init__(self, ctx)
        p = np.ascontiguousarray(params33, dtype=np.float64).reshape(33)
        _lib.check(_lib.load().djb_brdf_create_sgd_from_params(self.ctx._h, C.c_void_p(p.ctypes.data), C.byref(self._h)))
        return self

    def gaf(self, h, i, o):
        return self._mq(49, h, i, o)

    def g1(self, k):
        return self._mq(50, k)


class abc(brdf, _model_queries):
    """djb::abc(name): ABC BRDF with the published per-material parameters (dj_brdf.h:514-535, 3617-3668)."""

    def __init__(self, name: str, ctx=None):
        super().__init__(ctx)
        _lib.check(_lib.load().djb_brdf_create_abc(self.ctx._h, name.encode(), C.byref(self._h)))

    @classmethod
    def from_params(cls, params9, ctx=None):
        self = cls.__new__(cls)
        brdf.__init__(self, ctx)
        p = np.ascontiguousarray(params9, dtype=np.float64).reshape(9)
        _lib.check(_lib.load().djb_brdf_create_abc_from_params(self.ctx._h, C.c_void_p(p.ctypes.data), C.byref(self._h)))
        return self

    def gaf(self, h, i, o):
        return self._mq(49, h, i, o, width=1)


def _get_samples(b) -> np.ndarray:
    """merl::get_samples / utia::get_samples (dj_brdf.h:132, 143): the table as the reference holds it."""
    n = C.c_int64(0)
    _lib.check(_lib.load().djb_brdf_get_samples(b._h, None, C.c_int64(0), C.byref(n)))
    out = np.empty(n.value, np.float64)
    _lib.check(_lib.load().djb_brdf_get_samples(b._h, C.c_void_p(out.ctypes.data), n, C.byref(n)))
    return out


class merl(brdf):
    """djb::merl(filename) (dj_brdf.h:126-133, 963-983).  ``merl.from_table`` builds the same
    object from the file payload in memory (3*n doubles, planes R, G, B)."""

    def __init__(self, filename: str, ctx=None):
        super().__init__(ctx)
        _lib.check(_lib.load().djb_brdf_create_merl_from_file(self.ctx._h, filename.encode(), C.byref(self._h)))

    @classmethod
    def from_table(cls, table, ctx=None):
        self = cls.__new__(cls)
        brdf.__init__(self, ctx)
        t = np.ascontiguousarray(table, dtype=np.float64).reshape(-1)
        _lib.check(_lib.load().djb_brdf_create_merl_from_memory(
            self.ctx._h, C.c_void_p(t.ctypes.data), C.c_int64(t.size // 3), C.byref(self._h)))
        return self

    def get_samples(self) -> np.ndarray:
        return _get_samples(self)


class utia(brdf):
    """djb::utia(filename) (dj_brdf.h:136-146, 1039-1059)."""

    def __init__(self, filename: str, ctx=None):
        super().__init__(ctx)
        _lib.check(_lib.load().djb_brdf_create_utia_from_file(self.ctx._h, filename.encode(), C.byref(self._h)))

    @classmethod
    def from_table(cls, table, ctx=None):
        self = cls.__new__(cls)
        brdf.__init__(self, ctx)
        t = np.ascontiguousarray(table, dtype=np.float64).reshape(-1)
        if t.size != 3 * 288 * 288:
            raise exc(1, "djb_error: UTIA table must hold 3*288*288 doubles")
        _lib.check(_lib.load().djb_brdf_create_utia_from_memory(self.ctx._h, C.c_void_p(t.ctypes.data), C.byref(self._h)))
        return self

    def get_samples(self) -> np.ndarray:
        return _get_samples(self)


# --------------------------------------------------------------------------- user-defined BRDFs (dj_brdf.h:74-109)
def fit_query_dirs(resolution: int):
    """The (i, o) pairs at which ``tabular(src, resolution)`` evaluates its source, in the reference's call order
    (dj_brdf.h:2494, 2610): two [n,3] float32 arrays; a pair the reference's loop never reaches is NaN."""
    return _query_dirs("djb_fit_query_dirs", (C.c_int(resolution),))


def fit_aniso_query_dirs(elevation_res: int, azimuthal_res: int):
    """the same for ``tabular_anisotropic(src, elevation_res, azimuthal_res)`` (dj_brdf.h:2545, 2671)"""
    return _query_dirs("djb_fit_aniso_query_dirs", (C.c_int(elevation_res), C.c_int(azimuthal_res)))


def _query_dirs(fn, head):
    f = getattr(_lib.load(), fn)
    n = C.c_int64()
    _lib.check(f(*head, C.c_int64(0), None, None, C.byref(n)))
    qi, qo = _Vec(np.empty((n.value, 3), np.float32)), _Vec(np.empty((n.value, 3), np.float32))
    _lib.check(f(*head, n, C.byref(qi.view), C.byref(qo.view), None))
    return qi.keep, qo.keep


def _sample_source(src, qi, qo) -> np.ndarray:
    """src.eval at the valid query slots (one vectorised call, slot order kept) -> [n,3] float32, zeros elsewhere"""
    ok = ~np.isnan(qo[:, 0])
    rgb = np.zeros((qi.shape[0], 3), np.float32)
    rgb[ok] = np.asarray(src.eval(qi[ok], qo[ok]), np.float32).reshape(-1, 3)
    return rgb


class user_brdf(brdf):
    """A BRDF defined by the CALLER: the reference's extension point -- ``class brdf`` with its public constructor and
    ``eval`` as the one pure virtual (dj_brdf.h:74-109).  Derive and override ``eval(i, o, user_param=None)`` (host
    [n,3] float32 arrays in, [n,3] out; vectorised numpy is the natural form).  The other operators are the
    reference's base-class defaults (dj_brdf.h:795-845); ``tabular`` / ``tabular_anisotropic`` fit such an object by
    calling its eval() on the host at the fit's query directions and running the fit kernels on the samples."""

    def __init__(self, ctx=None):
        super().__init__(ctx)
        self._base = lambert(ctx=self.ctx)      # brdf::sample / brdf::pdf are what lambert inherits unchanged

    def eval(self, i, o, user_param=None):      # pure virtual
        raise NotImplementedError("djb_error: a user_brdf must override eval(i, o, user_param=None)")

    @staticmethod
    def _host(a):
        return np.ascontiguousarray(a, np.float32).reshape(-1, 3)

    def evalp(self, i, o, user_param=None):     # dj_brdf.h:803-806: eval * i.z
        i = self._host(i)
        return np.asarray(self.eval(i, self._host(o), user_param), np.float32).reshape(-1, 3) * i[:, 2:3]

    def eval_hd(self, h, d, user_param=None):   # dj_brdf.h:795-801
        i, o = brdf.hd_to_io(self._host(h), self._host(d), ctx=self.ctx)
        return np.asarray(self.eval(i, o, user_param), np.float32).reshape(-1, 3)

    def evalp_hd(self, h, d, user_param=None):  # dj_brdf.h:808-814
        i, o = brdf.hd_to_io(self._host(h), self._host(d), ctx=self.ctx)
        return np.asarray(self.eval(i, o, user_param), np.float32).reshape(-1, 3) * i[:, 2:3]

    def pdf(self, i, o, user_param=None):       # dj_brdf.h:842-845
        return self._base.pdf(self._host(i), self._host(o))

    def eval_pdf(self, i, o, user_param=None, cos=False):
        return (self.evalp if cos else self.eval)(i, o, user_param), self.pdf(i, o)

    def sample(self, u1, u2, o, user_param=None):   # dj_brdf.h:830-840
        return self._base.sample(u1, u2, self._host(o))

    def sample_rng(self, seed_u1, seed_u2, o, user_param=None, start=0):
        raise exc(1, "djb_error: sample_rng needs a BRDF resident on the GPU")

    def evalp_is(self, u1, u2, o, user_param=None):   # dj_brdf.h:816-828: evalp(i_, o) / pdf(i_, o), vec3 / float = (1.0 / b) * a
        o = self._host(o)
        i_ = self.sample(u1, u2, o, user_param)
        pdf_ = self.pdf(i_, o)
        with np.errstate(divide="ignore", invalid="ignore"):
            inv = (1.0 / pdf_.astype(np.float64)).astype(np.float32)
            w = inv[:, None] * self.evalp(i_, o, user_param)
        return w, i_, pdf_


class tabular(microfacet):
    """djb::tabular(brdf, res, shadow): the power-iteration fit, executed by the HIP fit kernel
    (dj_brdf.h:394-425, 2215-2236)."""

    def __init__(self, src: brdf, resolution: int, shadow: bool = True, ctx=None):
        brdf.__init__(self, ctx or src.ctx)
        if isinstance(src, user_brdf):      # host code: eval() sampled where the reference calls it (dj_brdf.h:2494, 2610)
            qi, qo = fit_query_dirs(resolution)
            self._from_samples(resolution, shadow, _sample_source(src, qi, qo))
        else:
            _lib.check(_lib.load().djb_brdf_create_tabular(self.ctx._h, src._h, C.c_int(resolution),
                                                           C.c_int(int(shadow)), C.byref(self._h)))
        self._fresnel = None

    def _from_samples(self, resolution, shadow, rgb):
        rgb = np.ascontiguousarray(rgb, np.float32).reshape(-1, 3)
        _lib.check(_lib.load().djb_brdf_create_tabular_from_samples(
            self.ctx._h, C.c_int(resolution), C.c_int(int(shadow)), C.c_void_p(rgb.ctypes.data), C.c_int64(rgb.shape[0]),
            C.byref(self._h)))

    @classmethod
    def from_samples(cls, resolution: int, rgb, shadow: bool = True, ctx=None):
        """the fit of a source known only through its values at ``fit_query_dirs(resolution)`` ([n,3] rgb; NaN slots ignored)"""
        t = cls.__new__(cls)
        brdf.__init__(t, ctx)
        t._from_samples(resolution, shadow, rgb)
        t._fresnel = None
        return t

    def supports_smith_vndf_sampling(self):
        return False

    def _get(self, which: int, width: int = 1):
        lib = _lib.load()
        n = C.c_int()
        _lib.check(lib.djb_tabular_get(self._h, C.c_int(which), None, C.byref(n)))
        a = np.empty((n.value, width) if width > 1 else (n.value,), dtype=np.float32)
        _lib.check(lib.djb_tabular_get(self._h, C.c_int(which), C.c_void_p(a.ctypes.data), None))
        return a

    def get_p22v(self):
        return self._get(0)

    def get_sigmav(self):
        return self._get(1)

    def get_cdfv(self):
        return self._get(2)

    def get_qfv(self):
        return self._get(3)

    def get_fresnel(self):
        if getattr(self, "_fresnel_replaced", False):
            return self._fresnel
        return fresnel.spline(self._get(4, 3))

    def _alphas(self):
        ab, ag = C.c_float(), C.c_float()
        _lib.check(_lib.load().djb_tabular_fit(self._h, C.byref(ab), C.byref(ag)))
        return ab.value, ag.value

    @staticmethod
    def fit_beckmann_parameters(tab: "tabular"):
        return microfacet.params.isotropic(tab._alphas()[0])

    @staticmethod
    def fit_ggx_parameters(tab: "tabular"):
        return microfacet.params.isotropic(tab._alphas()[1])


class tabular_anisotropic(microfacet):
    """djb::tabular_anisotropic(brdf, elevation_res, azimuthal_res, shadow) (dj_brdf.h:428-478):
    the anisotropic power-iteration fit, executed by djb_kernels_fit_aniso.hip."""

    def __init__(self, src: brdf, elevation_res: int, azimuthal_res: int, shadow: bool = True, ctx=None):
        brdf.__init__(self, ctx or src.ctx)
        if isinstance(src, user_brdf):      # dj_brdf.h:2545, 2671
            qi, qo = fit_aniso_query_dirs(elevation_res, azimuthal_res)
            self._from_samples(elevation_res, azimuthal_res, shadow, _sample_source(src, qi, qo))
        else:
            _lib.check(_lib.load().djb_brdf_create_tabular_anisotropic(
                self.ctx._h, src._h, C.c_int(elevation_res), C.c_int(azimuthal_res), C.c_int(int(shadow)),
                C.byref(self._h)))
        self._fresnel = None

    def _from_samples(self, elevation_res, azimuthal_res, shadow, rgb):
        rgb = np.ascontiguousarray(rgb, np.float32).reshape(-1, 3)
        _lib.check(_lib.load().djb_brdf_create_tabular_anisotropic_from_samples(
            self.ctx._h, C.c_int(elevation_res), C.c_int(azimuthal_res), C.c_int(int(shadow)), C.c_void_p(rgb.ctypes.data),
            C.c_int64(rgb.shape[0]), C.byref(self._h)))

    @classmethod
    def from_samples(cls, elevation_res: int, azimuthal_res: int, rgb, shadow: bool = True, ctx=None):
        """the fit of a source known only through its values at ``fit_aniso_query_dirs(elevation_res, azimuthal_res)``"""
        t = cls.__new__(cls)
        brdf.__init__(t, ctx)
        t._from_samples(elevation_res, azimuthal_res, shadow, rgb)
        t._fresnel = None
        return t

    def supports_smith_vndf_sampling(self):
        return False

    def _get(self, which: int, width: int = 1):
        lib = _lib.load()
        n, e, a = C.c_int(), C.c_int(), C.c_int()
        _lib.check(lib.djb_tabular_anisotropic_get(self._h, C.c_int(which), None, C.byref(n), C.byref(e), C.byref(a)))
        arr = np.empty((n.value, width) if width > 1 else (n.value,), dtype=np.float32)
        _lib.check(lib.djb_tabular_anisotropic_get(self._h, C.c_int(which), C.c_void_p(arr.ctypes.data), None, None, None))
        return arr, e.value, a.value

    def get_p22v(self):
        """(values, elev_cnt, azim_cnt), dj_brdf.h:447"""
        return self._get(0)

    def get_sigmav(self):
        return self._get(1)

    def get_table(self, name: str):
        return self._get({"pdf1": 2, "cdf1": 3, "qf1": 4, "pdf2": 5, "cdf2": 6, "qf2": 7}[name])[0]

    def qf2_entries(self) -> int:
        """size of the reference's m_qf2: elev*azim unless conditional-CDF rows came up short (dj_brdf.h:3005-3034)"""
        n = C.c_int()
        _lib.check(_lib.load().djb_tabular_anisotropic_get(self._h, C.c_int(9), None, C.byref(n), None, None))
        return n.value

    def get_fresnel(self):
        if getattr(self, "_fresnel_replaced", False):
            return self._fresnel
        return fresnel.spline(self._get(8, 3)[0])

    def _fits(self):
        b, g = _lib.Params(), _lib.Params()
        _lib.check(_lib.load().djb_tabular_anisotropic_fit(self._h, C.byref(b), C.byref(g)))
        return b, g

    @staticmethod
    def fit_beckmann_parameters(tab: "tabular_anisotropic"):
        b, _ = tab._fits()
        return microfacet.params(2, tuple(b.v))

    @staticmethod
    def fit_ggx_parameters(tab: "tabular_anisotropic"):
        _, g = tab._fits()
        return microfacet.params(2, tuple(g.v))

    # sampling queries (dj_brdf.h:450-455)
    def pdf1(self, phi):
        return self._query(32, self._cols(phi))

    def cdf1(self, phi):
        return self._query(33, self._cols(phi))

    def qf1(self, u1):
        return self._query(34, self._cols(u1))

    def pdf2(self, theta, phi):
        return self._query(35, self._cols(theta, phi))

    def cdf2(self, theta, phi):
        return self._query(36, self._cols(theta, phi))

    def qf2(self, u, phi):
        return self._query(37, self._cols(u, phi))


# --------------------------------------------------------------------------- batch fitter
def fit_merl_batch(tables, res: int = 90, shadow: bool = True, ctx: Optional[Context] = None,
                   return_tables: bool = False):
    """examples/merl_params.cpp:53-67 for many materials at once.

    ``tables``: sequence of float64 arrays (MERL payloads, 3*1458000 doubles each).
    Returns (alpha_beckmann[n], alpha_ggx[n]) and optionally the per-material tables."""
    lib = _lib.load()
    ctx = ctx or default_context()
    tabs = [np.ascontiguousarray(t, dtype=np.float64).reshape(-1) for t in tables]
    n = len(tabs)
    ptrs = (C.c_void_p * max(n, 1))(*[t.ctypes.data for t in tabs])
    ab, ag = np.zeros(n, np.float32), np.zeros(n, np.float32)
    extra = {}
    args = []
    for name, width in (("p22", 1), ("sigma", 1), ("cdf", 1), ("qf", 1), ("fresnel", 3)):
        if return_tables:
            extra[name] = np.zeros((n, res, width) if width > 1 else (n, res), np.float32)
            args.append(C.c_void_p(extra[name].ctypes.data))
        else:
            args.append(None)
    _lib.check(lib.djb_fit_merl_batch(ctx._h, C.c_int(n), ptrs, C.c_int(res), C.c_int(int(shadow)),
                                      C.c_void_p(ab.ctypes.data), C.c_void_p(ag.ctypes.data), *args))
    return (ab, ag, extra) if return_tables else (ab, ag)


def fit_brdf_batch(brdfs, res: int = 90, shadow: bool = True, ctx: Optional[Context] = None):
    """The same fit for BRDF objects already resident in HBM (no host->device traffic)."""
    lib = _lib.load()
    ctx = ctx or brdfs[0].ctx
    n = len(brdfs)
    ptrs = (C.c_void_p * max(n, 1))(*[b._h.value for b in brdfs])
    ab, ag = np.zeros(n, np.float32), np.zeros(n, np.float32)
    _lib.check(lib.djb_fit_brdf_batch(ctx._h, C.c_int(n), ptrs, C.c_int(res), C.c_int(int(shadow)),
                                      C.c_void_p(ab.ctypes.data), C.c_void_p(ag.ctypes.data),
                                      None, None, None, None, None))
    return ab, ag


# --------------------------------------------------------------------------- synthetic workloads (device)
def gen_directions(n: int, seed: int, start: int = 0, ctx: Optional[Context] = None, device=None):
    """[3, n] torch CUDA tensor of hash-generated unit vectors; same bits as synth.directions."""
    ctx = ctx or default_context()
    out = np.empty((3, n), np.float32) if ctx.is_cpu else torch.empty((3, n), dtype=torch.float32, device=device or f"cuda:{ctx.device}")
    v = _Vec(out)
    _lib.check(_lib.load().djb_gen_directions(ctx._h, C.c_int64(n), C.c_uint32(seed), C.c_uint64(start),
                                              C.byref(v.view)))
    return out


def gen_uniforms(n: int, seed: int, start: int = 0, ctx: Optional[Context] = None, device=None):
    ctx = ctx or default_context()
    if ctx.is_cpu:
        out = np.empty((n,), np.float32)
        _lib.check(_lib.load().djb_gen_uniforms(ctx._h, C.c_int64(n), C.c_uint32(seed), C.c_uint64(start), C.c_void_p(out.ctypes.data)))
        return out
    out = torch.empty((n,), dtype=torch.float32, device=device or f"cuda:{ctx.device}")
    _lib.check(_lib.load().djb_gen_uniforms(ctx._h, C.c_int64(n), C.c_uint32(seed), C.c_uint64(start),
                                            C.c_void_p(out.data_ptr())))
    return out


def histogram_xy(v, bins: int = 64, ctx: Optional[Context] = None):
    """bins x bins histogram of (x, y) in [-1, 1]^2 of a device vec3 batch (LDS atomics)."""
    ctx = ctx or default_context()
    vv = _Vec(v)
    counts = torch.zeros((bins * bins,), dtype=torch.int64, device=vv.device)
    _lib.check(_lib.load().djb_histogram_xy(ctx._h, C.c_int64(vv.n), C.byref(vv.view), C.c_int(bins),
                                            C.c_void_p(counts.data_ptr())))
    return counts.view(bins, bins)


# --------------------------------------------------------------------------- two-tier MERL diagnostics
def set_merl_exact_only(ctx: Context, on: bool):
    """Force merl eval/evalp onto the operation-by-operation fp64 kernel (DJB_OPT_MERL_EXACT_ONLY)."""
    _lib.check(_lib.load().djb_ctx_set_option(ctx._h, C.c_int(1), C.c_int(int(on))))


def set_fit_files_dense(ctx: Context, on: bool):
    """djb_fit_merl_files uploads and converts every table in full (DJB_OPT_FIT_FILES_DENSE) instead of fetching only the
    entries a tabular(merl, res) fit reads; same alphas."""
    _lib.check(_lib.load().djb_ctx_set_option(ctx._h, C.c_int(4), C.c_int(int(on))))


def set_scalar_on_device(ctx: Context, on: bool):
    """Send scalar-size host calls (<= DJB_SCALAR_HOST_MAX units) through the GPU as well (DJB_OPT_SCALAR_ON_DEVICE);
    by default the host instantiation of the same code answers them on the calling thread."""
    _lib.check(_lib.load().djb_ctx_set_option(ctx._h, C.c_int(3), C.c_int(int(on))))


def set_host_batch_max(ctx: Context, units: int):
    """Host-array calls of up to `units` units are answered on the calling thread by the object's host twin instead of a GPU round
    trip (DJB_OPT_HOST_BATCH_MAX; default DJB_SCALAR_HOST_MAX = 96).  Same bits either way."""
    _lib.check(_lib.load().djb_ctx_set_option(ctx._h, C.c_int(8), C.c_int(int(units))))


def set_utia_exact_only(ctx: Context, on: bool):
    """utia eval / evalp batches run the one-kernel form with the exact fall-backs inline (DJB_OPT_UTIA_EXACT_ONLY)
    instead of the two-tier kernel; same bits."""
    _lib.check(_lib.load().djb_ctx_set_option(ctx._h, C.c_int(5), C.c_int(int(on))))


def set_contract_1e5(ctx: Context, on: bool):
    """DJB_OPT_CONTRACT_1E5 (off by default): results inside the north star's VALUE contract -- within 1e-5 relative of the
    reference's, zeros / NaNs exactly where it has them -- instead of bit-identically, always as two tiers (fast fp32
    arithmetic + the bit-exact code for pairs whose decisions or conditioning are in doubt).  Reaches dense device-resident
    eval / evalp / pdf batches of GGX / Beckmann (ideal, Schlick, unpolarized Fresnel), ABC and SGD; `sample` of a Beckmann
    lobe (directions within 1e-5 per component); weight and pdf of `evalp_is` (GGX, Beckmann; the sampled direction stays
    the reference's, bit for bit).  Everything else stays bit-identical (include/djb_hip.h)."""
    _lib.check(_lib.load().djb_ctx_set_option(ctx._h, C.c_int(6), C.c_int(int(on))))


def selftest_contract(brdf, params=None, n: int = 1 << 24, seed: int = 1, family: int = 0, ctx: Optional[Context] = None):
    """The contract-mode fast path against the bit-exact per-pair code on n generated pairs (djb_selftest_contract)."""
    ctx = ctx or default_context()
    mx = (C.c_float * 2)()
    c = (C.c_ulonglong * 4)()
    _lib.check(_lib.load().djb_selftest_contract(ctx._h, brdf._h, C.byref(params._p) if params is not None else None,
                                                 C.c_int64(n), C.c_uint32(seed), C.c_int(family), mx, c))
    return {"max_rel_eval": float(mx[0]), "max_rel_pdf": float(mx[1]), "pairs": int(c[0]), "tier2": int(c[1]),
            "zero_mismatch": int(c[2]), "outside_1e5": int(c[3])}


def selftest_contract_sample(brdf, params=None, n: int = 1 << 24, seed: int = 1, family: int = 0, ctx: Optional[Context] = None):
    """The contract-mode sampler of a Beckmann or GGX lobe against the bit-exact per-sample code on n generated samples
    (djb_selftest_contract_sample): directions must agree to 1e-5 per component wherever the fast path keeps the sample."""
    ctx = ctx or default_context()
    mx = (C.c_float * 2)()
    c = (C.c_ulonglong * 4)()
    _lib.check(_lib.load().djb_selftest_contract_sample(ctx._h, brdf._h, C.byref(params._p) if params is not None else None,
                                                        C.c_int64(n), C.c_uint32(seed), C.c_int(family), mx, c))
    return {"max_abs_dir": float(mx[0]), "bound_used": float(mx[1]), "samples": int(c[0]), "exact_path": int(c[1]), "outside_1e5": int(c[2]),
            "degenerate_kept": int(c[3])}


def contract_sample_attack(brdf, u1, u2, o, params=None, iters: int = 256, seed: int = 1, ctx: Optional[Context] = None):
    """Directed search for the largest difference between the contract-mode sampler (Beckmann or GGX lobe) and the bit-exact code
    (djb_contract_sample_attack): u1, u2 ([n]) and o ([3, n]) are device tensors of candidates, hill-climbed IN PLACE over
    their bit patterns.  Returns (score per candidate in units of the contract, {evaluations, outside, accepted})."""
    ctx = ctx or default_context()
    vo = _Vec(o)
    best = torch.zeros((vo.n,), dtype=torch.float32, device=o.device)
    counters = (C.c_ulonglong * 3)()
    _lib.check(_lib.load().djb_contract_sample_attack(ctx._h, brdf._h, C.byref(params._p) if params is not None else None, C.c_int64(vo.n),
                                                      C.c_void_p(u1.data_ptr()), C.c_void_p(u2.data_ptr()), C.byref(vo.view), C.c_int(iters),
                                                      C.c_uint32(seed), C.c_void_p(best.data_ptr()), counters))
    return best, {"evaluations": int(counters[0]), "outside": int(counters[1]), "accepted": int(counters[2])}


def set_test_worklist_cap(ctx: Context, entries: int):
    """tests: override the tier-2 worklist capacity of the two-tier kernels (-1 = automatic); DJB_OPT_TEST_WORKLIST_CAP"""
    _lib.check(_lib.load().djb_ctx_set_option(ctx._h, C.c_int(7), C.c_int(int(entries))))


def set_aniso_qf2_aligned(ctx: Context, on: bool):
    """tabular_anisotropic objects built afterwards keep the rows of the conditional quantile table aligned
    (DJB_OPT_ANISO_QF2_ALIGNED) instead of reproducing the reference's shifted vector."""
    _lib.check(_lib.load().djb_ctx_set_option(ctx._h, C.c_int(2), C.c_int(int(on))))


def host_libm_status():
    """(matches, mode, atan_log_kat): does the host's libm return glibc 2.35's bits on the probe set (1 / 0 / -1 = not checked),
    which libm the host path calls (0 = the host's, 1 = the kernels' restatements compiled for the host), and whether the
    host's atan / log match their known answers (djb_ctx_libm_matches_host, djb_host_libm_mode, djb_host_atan_log_kat)."""
    lib = _lib.load()
    return int(lib.djb_ctx_libm_matches_host(None)), int(lib.djb_host_libm_mode()), int(lib.djb_host_atan_log_kat())


def selftest_fast_trig(n: int, mode: int, first: int = 0, seed: int = 1, ctx: Optional[Context] = None):
    """The arctangent core behind the trig sites of the table-driven kinds against the sites' previous forms (djb_selftest_fast_trig):
    mode 0 = the n floats after bit pattern `first` as utia's polar cosines, 1 / 8 = n generated (y, x) pairs (scale r2d / 1),
    2..7 = the n floats after `first` through one site with and without the core.
    Returns {decided, mismatch (must be 0), undecided, worst_ulp64 (a decided double's distance from the device libm's)}."""
    ctx = ctx or default_context()
    c = (C.c_ulonglong * 4)()
    _lib.check(_lib.load().djb_selftest_fast_trig(ctx._h, C.c_int64(n), C.c_int(mode), C.c_uint32(first), C.c_uint32(seed), c))
    return {"decided": c[0], "mismatch": c[1], "undecided": c[2], "worst_ulp64": int(c[3])}


def selftest_model_fast(brdf, n: int, seed: int = 1, first: int = 0, ctx: Optional[Context] = None):
    """The decided fast tier of an sgd / abc model's fp64 terms against the reference's chains (djb_selftest_model_fast): n generated
    polar cosines, or -- seed = 0 -- the n floats whose bit patterns follow `first` (first = 1, n = 0x3f800000: every float of (0, 1]).
    The `*_mismatch` counters must be 0."""
    ctx = ctx or brdf.ctx
    c = (C.c_ulonglong * 6)()
    _lib.check(_lib.load().djb_selftest_model_fast(ctx._h, brdf._h, C.c_int64(n), C.c_uint32(seed), C.c_uint32(first), c))
    return {"g1": c[0], "g1_undecided": c[1], "g1_mismatch": c[2], "ndf": c[3], "ndf_undecided": c[4], "ndf_mismatch": c[5]}


def selftest_guarded_math(n: int, seed: int = 1, ctx: Optional[Context] = None):
    """Self-test of the kernels' guarded fp64 shortcuts against the exact double sequences on n
    hash-generated inputs (see djb_selftest_guarded_math).  Mismatch counters must be 0."""
    ctx = ctx or default_context()
    c = (C.c_ulonglong * 12)()
    _lib.check(_lib.load().djb_selftest_guarded_math(ctx._h, C.c_int64(n), C.c_uint32(seed), c))
    return {"rsqrt_mismatch": c[0], "recip_mismatch": c[1], "rsqrt_fallback": c[2], "recip_fallback": c[3],
            "srgb_mismatch": c[4], "srgb_fallback": c[5], "fdiv_mismatch": c[6], "fdiv_fallback": c[7],
            "sqrt_mismatch": c[8], "sqrt_fallback": c[9], "div_mismatch": c[10], "div_fallback": c[11]}


def selftest_libm(fn: str, x, y=None, ctx: Optional[Context] = None):
    """The kernels' restatement of a host libm function (exp, pow, logf, expf, powf, atan2(x, y)), evaluated on the GPU
    (djb_selftest_libm); float functions take and return values representable as float."""
    ctx = ctx or default_context()
    code = {"exp": 0, "pow": 1, "logf": 2, "expf": 3, "powf": 4, "atan2": 5, "atan2_f32": 6, "atan2_deg_f32": 7, "sin": 8, "cos": 9, "tan": 10, "acos": 11}[fn]
    x = np.ascontiguousarray(x, np.float64).reshape(-1)
    y = np.ascontiguousarray(x if y is None else y, np.float64).reshape(-1)
    out = np.empty_like(x)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    _lib.check(_lib.load().djb_selftest_libm(ctx._h, C.c_int(code), C.c_int64(x.size), p(x), p(y), p(out)))
    return out


TRIG_SITES = ("cos", "sin", "tan", "acos", "acos_u", "acos_u32", "atan_squ", "atan_u", "atan_sqrt", "beck_qf", "acos_deg",
              "utia_bin15", "utia_bin7p5")
TRIG_DOUBLE_SITES = ("cos_d", "sin_d", "tan_d", "acos_d")


def _trig_code(fn):
    if isinstance(fn, str):
        return 16 + TRIG_DOUBLE_SITES.index(fn) if fn in TRIG_DOUBLE_SITES else TRIG_SITES.index(fn)
    return int(fn)


def selftest_trig_sweep(fn, first_bits: int, count: int, threads: int = 0, cap: int = 64, host_fn=None,
                        ctx: Optional[Context] = None):
    """Trig site `fn` (a name from TRIG_SITES / TRIG_DOUBLE_SITES or its code) on the GPU against the library's host
    instantiation (glibc) for the floats with bit patterns first_bits .. first_bits + count - 1
    (djb_selftest_trig_sweep).  Returns (number of differing inputs, rows): rows are (input bits, device bits, host
    bits), or (input bits, difference in ulps, 0) for a double site.  host_fn: another site on the host side
    (negative control)."""
    ctx = ctx or default_context()
    code = _trig_code(fn)
    n_bad = C.c_ulonglong(0)
    bad = np.zeros((max(cap, 1), 3), np.uint32)
    _lib.check(_lib.load().djb_selftest_trig_sweep(ctx._h, C.c_int(code), C.c_int(code if host_fn is None else _trig_code(host_fn)),
                                                   C.c_uint32(first_bits), C.c_int64(count), C.c_int(threads), C.byref(n_bad),
                                                   bad.ctypes.data_as(C.c_void_p), C.c_int(cap)))
    k = min(int(n_bad.value), cap)
    return int(n_bad.value), [tuple(int(v) for v in row) for row in bad[:k]]


def _guard6(guard):
    """{a_h, b_h, a_d, b_d, c_d, a_p} in units of 2^-24; a 5-tuple (the round-1/2 form) uses a_p = a_d"""
    g = list(guard)
    if len(g) == 5:
        g.append(g[2])
    return (C.c_float * 6)(*g)


def merl_guard_stats(i, o, guard=None, ctx: Optional[Context] = None):
    """Calibration of the two-tier MERL kernel on device-resident pairs (see djb_merl_guard_stats)."""
    ctx = ctx or default_context()
    vi, vo = _Vec(i), _Vec(o)
    ratios = (C.c_float * 3)()
    counters = (C.c_ulonglong * 4)()
    g = None
    if guard is not None:
        g = _guard6(guard)
    _lib.check(_lib.load().djb_merl_guard_stats(ctx._h, C.c_int64(vi.n), C.byref(vi.view), C.byref(vo.view), g,
                                                ratios, counters))
    return {"max_ratio": tuple(ratios), "special": counters[0], "ambiguous": counters[1],
            "mismatch": counters[2], "certain": counters[3]}


def merl_guard_attack(i, o, iters: int = 256, seed: int = 1, guard=None, ctx: Optional[Context] = None):
    """Directed search for the worst |tier-1 estimate - reference| / guard band (djb_merl_guard_attack): i, o are
    [3, n] device tensors of candidate pairs, hill-climbed IN PLACE over their bit patterns.  Returns
    (best ratio per candidate as a device tensor, {evaluations, mismatch, accepted})."""
    ctx = ctx or default_context()
    vi, vo = _Vec(i), _Vec(o)
    best = torch.zeros((vi.n,), dtype=torch.float32, device=i.device)
    counters = (C.c_ulonglong * 3)()
    g = _guard6(guard) if guard is not None else None
    _lib.check(_lib.load().djb_merl_guard_attack(ctx._h, C.c_int64(vi.n), C.byref(vi.view), C.byref(vo.view), g, C.c_int(iters),
                                                 C.c_uint32(seed), C.c_void_p(best.data_ptr()), counters))
    return best, {"evaluations": int(counters[0]), "mismatch": int(counters[1]), "accepted": int(counters[2])}
