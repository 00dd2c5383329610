#!/usr/bin/env python3
"""bench.py -- throughput of the dj_brdf hot path on MI355X, one JSON line on rank 0.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--n PAIRS]

A "step" is one pass of the hot path over one batch of synthetic input that is already resident
in HBM (directions generated on-device by the counter hash of dj_brdf_amd/synth.py).

Workloads (BASELINE.json configs):
  merl_eval        configs[2]: MERL tabulated eval (nearest-bin, bit-exact index) over 1e9 (wi,wo)
                   pairs -- the configuration BASELINE.json's north_star quotes its target on
                   (>= 1 G evals/s on one MI355X); DEFAULT.
  ggx_eval_pdf     configs[1]: GGX isotropic conductor eval()+pdf() fused, 1e8 pairs, alpha = 0.3
  beckmann_sample  configs[3]: Beckmann elliptic(0.2,0.5,0.7) VNDF sample(), 1e9 samples, on-chip RNG
  merl_fit         configs[4]: power-iteration fit of 100 MERL materials resident in HBM
  utia_eval        (no BASELINE config; north_star names UTIA tables) utia::eval over 1e8 pairs
  ggx_eval_pdf_contract   configs[1] with DJB_OPT_CONTRACT_1E5 (values within 1e-5 relative, not bit-identical)
  merl_eval_uniform_bins  configs[2] with look-ups spread uniformly over all 1.458 M bins (worst case for the caches)
  merl_eval_coherent      configs[2] on a renderer-like batch: neighbouring pixels of a bumpy plane, one light

Multi-GPU (torchrun, one rank per GPU): units are independent, every rank runs the same per-GPU
batch on its own device ("weak" scaling), there is NO data-path collective; the only
communication is the barrier and the MAX-over-ranks of the timed region.

Extra objects in the JSON line:
  secondary     (default workload only) BASELINE.json's second figure, the 100-material MERL fit on N GPUs
                (material m -> rank m mod N, "strong", no exchange):
                  merl_fit_files_100  END TO END, what examples/merl_params.cpp:53-69 does: 100 files of
                                      34 992 012 B on local disk -> pread -> PCIe -> k_merl_convert -> one fit
                                      launch -> alphas, wall time = max over ranks, with the load / fit split
                                      and (N=1) the reference's own merl_params binary timed on a few of the files;
                  merl_fit_100        compute only (tables already resident in HBM);
                at N=1 also the other single-GPU configs (ggx_eval_pdf, beckmann_sample) and utia_eval;
                  one_pair_calls      ns per ONE call of the facade's virtuals (what a renderer issues): a GPU object answered by
                                      its host twin, the CPU context, and (cpu_baseline inside it) the real reference built from
                                      the same source (examples/scalar_latency.cpp, oracle/_ref/scalar_latency)
  roofline      dominant kernel: algorithmic bytes per launch / average launch duration
                (HIP events on the ctx stream over the timed region) vs the 8 TB/s HBM peak
  cpu_baseline  the CPU path timed on this host (rank 0, N=1 only) on a bounded sample:
                kind "reference" = the real dj_brdf.h via oracle/_ref (prebuilt), else
                "port" = oracle/djb_oracle.c
"""
import argparse
import ctypes
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)

WORKLOADS = {
    # name: (default n per GPU, algorithmic HBM bytes per unit, unit, kernel family)
    "merl_eval": (1_000_000_000, 36, "evals", "k_merl_fast_v4<eval> (two-tier exact, tier 2 drained in-kernel)"),
    "ggx_eval_pdf": (100_000_000, 40, "evals", "k_eval<GGX,eval+pdf>"),
    # the same configuration under DJB_OPT_CONTRACT_1E5: values within 1e-5 relative instead of bit-identical
    "ggx_eval_pdf_contract": (100_000_000, 40, "evals", "k_ct_fast_v4<GGX,eval+pdf> + k_ct_fixup (two-tier, 1e-5 value contract)"),
    # the headline kernel on two other look-up distributions (the 0.40 of merl_eval is distribution dependent):
    "merl_eval_uniform_bins": (1_000_000_000, 36, "evals", "k_merl_fast_v4<eval>, (theta_h, theta_d, phi_d) bins uniform over the table"),
    "merl_eval_coherent": (1_000_000_000, 36, "evals", "k_merl_fast_v4<eval>, renderer-like batch (neighbouring pixels of a bumpy plane)"),
    # kinds the round-4 contract mode reaches (no BASELINE config of their own): exact kernel and DJB_OPT_CONTRACT_1E5
    "ggx_unpolarized_eval_pdf": (100_000_000, 40, "evals", "k_eval<GGX,eval+pdf,unpolarized>"),
    "ggx_unpolarized_eval_pdf_contract": (100_000_000, 40, "evals", "k_ct_fast_v4<GGX,eval+pdf,unpolarized> + k_ct_fixup (1e-5 value contract)"),
    "sgd_eval": (100_000_000, 36, "evals", "k_eval<SGD,eval> (gold-metallic-paint)"),
    "sgd_eval_contract": (100_000_000, 36, "evals", "k_ct_fast_v4<SGD,eval> + k_ct_fixup (gold-metallic-paint, 1e-5 value contract)"),
    "beckmann_sample": (1_000_000_000, 24, "samples", "k_sample_bk<sample,rng>"),
    # configs[3] under DJB_OPT_CONTRACT_1E5: every component of the sampled direction within 1e-5 of the reference's
    "beckmann_sample_contract": (1_000_000_000, 24, "samples", "k_sample_bk<sample,rng,contract> (fp32 Newton sequence that follows the reference's; doubtful samples re-done exactly in the same launch)"),
    "utia_eval": (100_000_000, 36, "evals", "k_utia_v2<eval> + k_eval_utia_fix<eval> (two-tier exact)"),
    # under DJB_OPT_CONTRACT_1E5: cells, weights and the 16-tap sums stay the reference's bits, the sRGB power runs on v_log / v_exp_f32
    "utia_eval_contract": (100_000_000, 36, "evals", "k_utia_v2<eval,contract> + k_eval_utia_fix<eval> (1e-5 value contract)"),
    # ---- the operators the five Mitsuba plugins issue that have no BASELINE config of their own: driver-run legs
    # (secondary.plugin_ops at N=1), each also a --workload of its own so that tools/profile_bench.sh can take its counters
    # dj_sgd / dj_abc: pdf() and sample() come from tabular(model, 90) (mitsuba/dj_abc.cpp:28-32, 77, 89); eval + pdf fused, 40 B
    "tabular_eval_pdf": (100_000_000, 40, "evals", "k_eval<TABULAR,eval+pdf> on tabular(ggx, 90)"),
    "tabular_sample": (100_000_000, 24, "samples", "k_sample<TABULAR,sample,rng> on tabular(ggx, 90) (normal-map scheme, dj_brdf.h:1806-1846)"),
    "tabular_abc_sample": (100_000_000, 24, "samples", "k_sample<TABULAR,sample,rng> on tabular(abc gold-metallic-paint, 90)"),
    # evalp_is (dj_brdf.h:1734-1765): u1, u2 (8) + o (12) -> weight (12) + i (12) + pdf (4) = 48 B with the uniforms read from HBM
    "ggx_evalp_is": (100_000_000, 48, "samples", "k_sample<GGX,evalp_is> elliptic(0.2, 0.5, 0.7)"),
    "beckmann_evalp_is": (100_000_000, 48, "samples", "k_sample_bk<evalp_is> elliptic(0.2, 0.5, 0.7)"),
    # dj_beckmannconductor's per-hit LEAN path (mitsuba/dj_beckmannconductor.cpp:291-319): i, o (24) + 5 moments (20) -> evalp (12) + pdf (4)
    "lean_evalp_pdf": (100_000_000, 60, "evals", "k_eval_pp<BECKMANN,lean,evalp+pdf> Schlick Fresnel, base isotropic(0.1)"),
    "abc_evalp": (100_000_000, 36, "evals", "k_eval<ABC,evalp> gold-metallic-paint (dj_abc::eval)"),
    "tabular_aniso_eval_pdf": (100_000_000, 40, "evals", "k_eval<TABULAR_ANISO,eval+pdf> on tabular_anisotropic(utia, 90, 90)"),
    "tabular_aniso_sample": (100_000_000, 24, "samples", "k_sample<TABULAR_ANISO,sample,rng> on tabular_anisotropic(utia, 90, 90)"),
    # the two fits as timed workloads of their own: one object construction per step (the table of the source resident in HBM)
    "fit_tabular_90": (1, None, "fits", "k_fit<MERL>: tabular(merl, 90) + both moment fits, ONE material"),
    "fit_aniso_90x90": (1, None, "fits", "launch_fit_aniso (18 launches): tabular_anisotropic(utia, 90, 90)"),
    "merl_fit": (100, None, "materials", "k_fit<MERL>"),
    # end to end: 100 MERL files (34 992 012 B each) on local disk -> params: pread + PCIe + convert + fit
    "merl_fit_files": (100, None, "materials", "djb_fit_merl_files (reader threads -> pinned ring -> H2D -> k_merl_convert -> k_fit<MERL>)"),
}


def synth_merl_files(n, synth, rank=0, distinct=10, only=None):
    """n MERL-format files named after the MERL materials; `distinct` different tables, repeated.
    only: indices to create (a rank's share); the returned list then holds just those paths."""
    d = f"/tmp/djb_bench_merl_r{rank}"
    os.makedirs(d, exist_ok=True)
    paths, blobs = [], {}
    for k in (range(n) if only is None else only):
        p = os.path.join(d, synth.MERL_NAMES[k % 100] + (f"_{k // 100}" if k >= 100 else "") + ".binary")
        if not (os.path.exists(p) and os.path.getsize(p) == synth.MERL_FILE_BYTES):
            r = k % distinct
            if r not in blobs:
                blobs[r] = synth.merl_table(*synth.material_recipe(r))
            synth.write_merl_binary(p, blobs[r])
        paths.append(p)
    return paths


# the kernel function(s) a workload launches in its timed region: a tracked profile summary (profiles/pmc_*.json, valu_*.json) is
# attached to the bench line only if it was taken on exactly these -- a summary that lists anything else describes a kernel
# that no longer exists (round 4 shipped a valu_merl_eval.json of the pre-fusion kernel pair) and is reported as stale instead
LAUNCHES = {
    "merl_eval": ["k_merl_fast_v4"], "merl_eval_uniform_bins": ["k_merl_fast_v4"], "merl_eval_coherent": ["k_merl_fast_v4"],
    "ggx_eval_pdf": ["k_eval"], "ggx_unpolarized_eval_pdf": ["k_eval"], "sgd_eval": ["k_eval"],
    "ggx_eval_pdf_contract": ["k_ct_fast_v4", "k_ct_fixup"], "ggx_unpolarized_eval_pdf_contract": ["k_ct_fast_v4", "k_ct_fixup"],
    "sgd_eval_contract": ["k_ct_fast_v4", "k_ct_fixup"],
    "beckmann_sample": ["k_sample_bk"], "beckmann_sample_contract": ["k_sample_bk"],
    "utia_eval": ["k_utia_v2", "k_eval_utia_fix"], "utia_eval_contract": ["k_utia_v2", "k_eval_utia_fix"], "merl_fit": ["k_fit"],
    "tabular_eval_pdf": ["k_eval"], "tabular_aniso_eval_pdf": ["k_eval"], "abc_evalp": ["k_eval"], "lean_evalp_pdf": ["k_eval_pp"],
    "tabular_sample": ["k_sample"], "tabular_abc_sample": ["k_sample"], "tabular_aniso_sample": ["k_sample"],
    "ggx_evalp_is": ["k_sample"], "beckmann_evalp_is": ["k_sample_bk"], "fit_tabular_90": ["k_fit"],
}
# kernels a workload launches besides (set-up, input generation): never part of a profile summary
NOT_THE_LEG = ("k_gen_dir", "k_gen_uni", "k_merl_convert", "k_utia_convert", "k_fit_smith_nint", "k_fit_fresnel_dirs", "k_fit_merl_slots")


def kernel_short_name(k):
    """`void djbk::(anonymous namespace)::k_eval<1, 5, 0>(...)` -> `k_eval`"""
    import re
    m = re.search(r"\b(ka?_[A-Za-z0-9_]+)", k.replace("(anonymous namespace)::", ""))
    return m.group(1) if m else None


def profile_matches(name, kernel_names):
    """(ok, listed): do the kernels a profile summary lists equal the set this workload launches?"""
    listed = sorted({kernel_short_name(k) for k in kernel_names if kernel_short_name(k)})
    want = sorted(LAUNCHES.get(name, []))
    return bool(want) and listed == want, listed


def drop_page_cache(paths):
    """Ask the kernel to drop the page cache of these files (fsync + posix_fadvise(DONTNEED): works for clean pages without root;
    a tmpfs keeps its pages, which is reported).  Returns the resident fraction afterwards, from mincore() on a sample of the files."""
    import ctypes
    libc = ctypes.CDLL("libc.so.6", use_errno=True)
    libc.mmap.restype = ctypes.c_void_p
    libc.mmap.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_long]
    libc.munmap.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    libc.mincore.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    for p in paths:
        fd = os.open(p, os.O_RDONLY)
        try:
            os.fsync(fd)
        except OSError:
            pass
        os.posix_fadvise(fd, 0, 0, os.POSIX_FADV_DONTNEED)
        os.close(fd)
    resident = total = 0
    page = os.sysconf("SC_PAGE_SIZE")
    for p in paths[:: max(1, len(paths) // 8)]:
        size = os.path.getsize(p)
        fd = os.open(p, os.O_RDONLY)
        addr = libc.mmap(None, size, 1, 1, fd, 0)                 # PROT_READ, MAP_SHARED
        os.close(fd)
        if addr in (None, ctypes.c_void_p(-1).value):
            continue
        pages = (size + page - 1) // page
        vec = (ctypes.c_ubyte * pages)()
        if libc.mincore(addr, size, vec) == 0:
            resident += sum(b & 1 for b in vec); total += pages
        libc.munmap(addr, size)
    return (resident / total) if total else None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="merl_eval", choices=list(WORKLOADS))
    ap.add_argument("--n", type=int, default=None, help="units per GPU per step (default: the BASELINE config size)")
    ap.add_argument("--alpha", type=float, default=0.3, help="ggx_eval_pdf: isotropic roughness (SURVEY 8d extra points 0.05 / 0.8)")
    ap.add_argument("--fresnel", default="ideal", choices=["ideal", "schlick"], help="ggx_eval_pdf: ideal (default) or schlick(1.0, 0.71, 0.29)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--merl-dir", default=None,
                    help="a directory of real MERL files (*.binary): the end-to-end files -> alphas leg also runs on them "
                         "(secondary.merl_fit_dir; examples/merl_params.cpp:53-69 is the loop it stands for).  The offline image has none.")
    ap.add_argument("--selftest-n", type=int, default=None,
                    help="harness self-test only (DJB_BENCH_SHARE_GPU runs of the N-rank path on one GPU): shrink the primary batch "
                         "to this many units per rank but keep the secondary fit legs; the line is labelled")
    return ap.parse_args()


GGX_ALPHA, GGX_FRESNEL = 0.3, "ideal"       # overridden by --alpha / --fresnel


def merl_pairs(name, n, djb, torch, ctx):
    """(i, o) as [3, n] device tensors for the two extra MERL legs, generated on the device in chunks.
    uniform_bins: a (theta_h, theta_d, phi_d) bin drawn uniformly from the 90 x 90 x 180 table, the pair at a jittered
    position inside it (hd_to_io of the bin's angles) -- every table line equally likely, nothing for the caches to keep.
    coherent: what a renderer hands over -- pixel k of a W-wide image of a gently bumpy plane, camera and one light
    fixed, i / o = light / view direction in the pixel's shading frame: consecutive pairs land in the same or
    neighbouring bins."""
    dev = f"cuda:{ctx.device}"
    i = torch.empty((3, n), dtype=torch.float32, device=dev)
    o = torch.empty((3, n), dtype=torch.float32, device=dev)
    CH = 1 << 26
    g = torch.Generator(device=dev); g.manual_seed(1234)
    for lo in range(0, n, CH):
        m = min(CH, n - lo)
        if name == "merl_eval_uniform_bins":
            u = torch.rand((3, m), generator=g, device=dev, dtype=torch.float32)
            ih = torch.randint(0, 90, (m,), generator=g, device=dev); idd = torch.randint(0, 90, (m,), generator=g, device=dev)
            ip = torch.randint(0, 180, (m,), generator=g, device=dev)
            j = 0.1 + 0.8 * u                                           # stay clear of the bin edges
            th = ((ih + j[0]) ** 2 / 90.0) * (3.14159265 / 180.0)       # theta_h bins are quadratic (dj_brdf.h:906-920)
            td = (idd + j[1]) * (3.14159265 / 180.0)
            pd = (ip + j[2]) * (3.14159265 / 180.0)
            h = torch.stack([torch.sin(th), torch.zeros_like(th), torch.cos(th)])
            d = torch.stack([torch.sin(td) * torch.cos(pd), torch.sin(td) * torch.sin(pd), torch.cos(td)])
            ii, oo = djb.brdf.hd_to_io(h.contiguous(), d.contiguous(), ctx=ctx)
            i[:, lo:lo + m] = ii; o[:, lo:lo + m] = oo
            del u, ih, idd, ip, j, th, td, pd, h, d, ii, oo
        else:
            W = 32768
            k = torch.arange(lo, lo + m, device=dev, dtype=torch.int64)
            x = (k % W).to(torch.float32) * (8.0 / W) - 4.0             # the plane spans [-4, 4] x [-4 H/W, ...]
            y = (k // W).to(torch.float32) * (8.0 / W) - 4.0
            # height field: three sinusoids, slopes <= ~0.45
            sx = 0.20 * torch.cos(3.1 * x + 0.5 * y) * 3.1 * 0.05 + 0.15 * torch.cos(0.7 * x - 1.3 * y) * 0.7 + 0.1 * torch.sin(5.0 * x)
            sy = 0.20 * torch.cos(3.1 * x + 0.5 * y) * 0.5 * 0.05 - 0.15 * torch.cos(0.7 * x - 1.3 * y) * 1.3 + 0.1 * torch.cos(4.0 * y)
            nrm = torch.rsqrt(sx * sx + sy * sy + 1.0)
            nx, ny, nz = -sx * nrm, -sy * nrm, nrm
            # tangent frame: t = normalize(e_x - n (n.e_x)), b = n x t
            tx, ty, tz = 1.0 - nx * nx, -ny * nx, -nz * nx
            tn = torch.rsqrt(tx * tx + ty * ty + tz * tz); tx, ty, tz = tx * tn, ty * tn, tz * tn
            bx, by, bz = ny * tz - nz * ty, nz * tx - nx * tz, nx * ty - ny * tx
            def local(vx, vy, vz):
                vn = torch.rsqrt(vx * vx + vy * vy + vz * vz); vx, vy, vz = vx * vn, vy * vn, vz * vn
                return torch.stack([vx * tx + vy * ty + vz * tz, vx * bx + vy * by + vz * bz, vx * nx + vy * ny + vz * nz])
            o[:, lo:lo + m] = local(0.0 - x, -6.0 - y, 5.0 + 0 * x)      # camera at (0, -6, 5)
            i[:, lo:lo + m] = local(3.0 - x, 2.0 - y, 6.0 + 0 * x)       # point light at (3, 2, 6)
            del k, x, y, sx, sy, nrm, nx, ny, nz, tx, ty, tz, tn, bx, by, bz
    torch.cuda.synchronize()
    return i, o


def contract_mode(step, name, djb, ctx):
    """A `*_contract` workload runs under DJB_OPT_CONTRACT_1E5.  The option is per context and off by default; it is switched ON
    ONCE here, outside every timed region, and OFF by step.cleanup() (finish()) when the leg is over -- toggling it around every
    launch would reset the context's per-lobe tier-2 statistics each time, so the timed loop would never run the code path a
    caller who simply leaves the option on gets."""
    if name.endswith("_contract"):
        djb.set_contract_1e5(ctx, True)
        step.cleanup = lambda: djb.set_contract_1e5(ctx, False)
    return step


def utia_contract_accuracy(step, keep, djb, ctx, torch):
    """The contract launch against the bit-exact launch on the leg's own pairs (all of them): the option only changes the sRGB power."""
    out = keep[3]
    try:
        djb.set_contract_1e5(ctx, True)           # the leg is over (finish() has switched the option off): on for one launch, off again
        step(); torch.cuda.synchronize()
        fast = out.clone()
    finally:
        djb.set_contract_1e5(ctx, False)
    step(); torch.cuda.synchronize()
    exact = out
    rel = (fast - exact).abs() / exact.abs().clamp_min(1e-30)
    rel = torch.where(exact == fast, torch.zeros_like(rel), rel)
    res = {"max_rel_err_eval": float(rel.max()), "values_outside_1e-5": int((rel > 1e-5).sum()),
           "zero_pattern_mismatches": int(((exact == 0) != (fast == 0)).sum()), "values_compared": int(exact.numel()),
           "bit_identical_share": float((exact.view(torch.int32) == fast.view(torch.int32)).float().mean()),
           "contract": "1e-5 relative (north_star); cells, weights and 16-tap sums are the reference's bits, only the sRGB power is "
                       "approximate; DJB_OPT_CONTRACT_1E5, off by default"}
    del fast, rel
    return res


def finish(step):
    getattr(step, "cleanup", lambda: None)()


def make_step(name, n, djb, synth, ctx, torch):
    """Returns (step_fn, keepalive).  Inputs are generated on-device before the timed region."""
    if name == "merl_eval":
        i = djb.gen_directions(n, synth.SEED_I, ctx=ctx)
        o = djb.gen_directions(n, synth.SEED_O, ctx=ctx)
        m = djb.merl.from_table(synth.merl_table(0.3), ctx=ctx)   # GGX 0.3 + diffuse at bin centres
        out = torch.empty((3, n), dtype=torch.float32, device=i.device)
        lib, C = djb._lib.load(), ctypes
        vi, vo, vout = djb._Vec(i), djb._Vec(o), djb._Vec(out)

        def step():
            djb._lib.check(lib.djb_eval_batch(ctx._h, m._h, C.c_int64(n), C.byref(vi.view), C.byref(vo.view),
                                              None, C.byref(vout.view), C.c_int(0)))
        return step, (i, o, m, out, vi, vo, vout)
    if name in ("merl_eval_uniform_bins", "merl_eval_coherent"):
        i, o = merl_pairs(name, n, djb, torch, ctx)
        m = djb.merl.from_table(synth.merl_table(0.3), ctx=ctx)
        out = torch.empty((3, n), dtype=torch.float32, device=i.device)
        lib, C = djb._lib.load(), ctypes
        vi, vo, vout = djb._Vec(i), djb._Vec(o), djb._Vec(out)

        def step():
            djb._lib.check(lib.djb_eval_batch(ctx._h, m._h, C.c_int64(n), C.byref(vi.view), C.byref(vo.view),
                                              None, C.byref(vout.view), C.c_int(0)))
        return step, (i, o, m, out, vi, vo, vout)
    if name in ("sgd_eval", "sgd_eval_contract"):
        i = djb.gen_directions(n, synth.SEED_I, ctx=ctx)
        o = djb.gen_directions(n, synth.SEED_O, ctx=ctx)
        m = djb.sgd("gold-metallic-paint", ctx=ctx)
        out = torch.empty((3, n), dtype=torch.float32, device=i.device)
        lib, C = djb._lib.load(), ctypes
        vi, vo, vout = djb._Vec(i), djb._Vec(o), djb._Vec(out)
        def step():
            djb._lib.check(lib.djb_eval_batch(ctx._h, m._h, C.c_int64(n), C.byref(vi.view), C.byref(vo.view),
                                              None, C.byref(vout.view), C.c_int(0)))
        return contract_mode(step, name, djb, ctx), (i, o, m, None, out, vi, vo, vout)
    if name in ("ggx_eval_pdf", "ggx_eval_pdf_contract", "ggx_unpolarized_eval_pdf", "ggx_unpolarized_eval_pdf_contract"):
        i = djb.gen_directions(n, synth.SEED_I, ctx=ctx)
        o = djb.gen_directions(n, synth.SEED_O, ctx=ctx)
        fr = djb.fresnel.unpolarized((1.5, 1.8, 2.4)) if "unpolarized" in name else \
            djb.fresnel.ideal() if GGX_FRESNEL == "ideal" else djb.fresnel.schlick((1.0, 0.71, 0.29))
        g = djb.ggx(fr, True, ctx=ctx)
        p = djb.microfacet.params.isotropic(GGX_ALPHA)
        out = torch.empty((3, n), dtype=torch.float32, device=i.device)
        pdf = torch.empty((n,), dtype=torch.float32, device=i.device)
        lib, C = djb._lib.load(), ctypes
        vi, vo, vout = djb._Vec(i), djb._Vec(o), djb._Vec(out)

        def step():
            djb._lib.check(lib.djb_eval_pdf_batch(ctx._h, g._h, C.c_int64(n), C.byref(vi.view), C.byref(vo.view),
                                                  C.byref(p._p), C.c_int(0), C.byref(vout.view),
                                                  C.c_void_p(pdf.data_ptr()), C.c_int(0)))
        return contract_mode(step, name, djb, ctx), (i, o, g, p, out, pdf, vi, vo, vout)
    if name in ("beckmann_sample", "beckmann_sample_contract"):
        o = djb.gen_directions(n, synth.SEED_O, ctx=ctx)
        b = djb.beckmann(djb.fresnel.ideal(), True, ctx=ctx)
        p = djb.microfacet.params.elliptic(0.2, 0.5, 0.7)
        out = torch.empty((3, n), dtype=torch.float32, device=o.device)
        lib, C = djb._lib.load(), ctypes
        vo, vout = djb._Vec(o), djb._Vec(out)

        def step():
            djb._lib.check(lib.djb_sample_rng_batch(ctx._h, b._h, C.c_int64(n), C.c_uint32(synth.SEED_U1),
                                                    C.c_uint32(synth.SEED_U2), C.c_uint64(0), C.byref(vo.view),
                                                    C.byref(p._p), C.byref(vout.view)))
        return contract_mode(step, name, djb, ctx), (o, out, b, p, vo, vout)
    if name in ("utia_eval", "utia_eval_contract"):
        i = djb.gen_directions(n, synth.SEED_I, ctx=ctx)
        o = djb.gen_directions(n, synth.SEED_O, ctx=ctx)
        tab = np.random.default_rng(11).uniform(0.0, 120.0, size=3 * 288 * 288)   # UTIA-format payload (sRGB-coded * 140)
        u = djb.utia.from_table(tab, ctx=ctx)
        out = torch.empty((3, n), dtype=torch.float32, device=i.device)
        lib, C = djb._lib.load(), ctypes
        vi, vo, vout = djb._Vec(i), djb._Vec(o), djb._Vec(out)

        def step():
            djb._lib.check(lib.djb_eval_batch(ctx._h, u._h, C.c_int64(n), C.byref(vi.view), C.byref(vo.view),
                                              None, C.byref(vout.view), C.c_int(0)))
        return contract_mode(step, name, djb, ctx), (i, o, u, out, vi, vo, vout)
    if name in PLUGIN_LEGS:
        return plugin_leg_step(name, n, djb, synth, ctx, torch)
    if name == "merl_fit":
        which = range(n) if isinstance(n, int) else n          # an explicit list of material indices (sharded fit)
        mats = [djb.merl.from_table(synth.merl_table(*synth.material_recipe(k)), ctx=ctx) for k in which]
        result = {}

        def step():
            result["alphas"] = djb.fit_brdf_batch(mats, 90, True, ctx=ctx)
        return step, (mats, result)
    if name == "merl_fit_files":
        from dj_brdf_amd import merl_params
        paths = synth_merl_files(n, synth, rank=int(os.environ.get("RANK", "0")))
        result = {}

        def step():
            ab, ag, result["timing"] = merl_params.fit_files_on(ctx, paths)
        return step, (paths, result)
    raise ValueError(name)


PLUGIN_LEGS = ("tabular_eval_pdf", "tabular_sample", "tabular_abc_sample", "ggx_evalp_is", "beckmann_evalp_is", "lean_evalp_pdf",
               "abc_evalp", "tabular_aniso_eval_pdf", "tabular_aniso_sample", "fit_tabular_90", "fit_aniso_90x90")


def utia_payload():
    return np.random.default_rng(11).uniform(0.0, 120.0, size=3 * 288 * 288)     # UTIA-format payload (sRGB-coded * 140)


def plugin_leg_step(name, n, djb, synth, ctx, torch):
    """The operators the Mitsuba plugins issue beyond the BASELINE configs, one launch (family) per step, inputs resident in HBM."""
    lib, C = djb._lib.load(), ctypes
    chk = djb._lib.check
    dev = f"cuda:{ctx.device}"
    if name == "fit_tabular_90":
        src = djb.merl.from_table(synth.merl_table(*synth.material_recipe(0)), ctx=ctx)
        def step():
            djb.tabular(src, 90, True, ctx=ctx).close()
        return step, (src,)
    if name == "fit_aniso_90x90":
        src = djb.utia.from_table(utia_payload(), ctx=ctx)
        def step():
            djb.tabular_anisotropic(src, 90, 90, True, ctx=ctx).close()
        return step, (src,)
    o = djb.gen_directions(n, synth.SEED_O, ctx=ctx)
    vo = djb._Vec(o)
    if name.startswith("tabular_aniso"):
        src = djb.utia.from_table(utia_payload(), ctx=ctx)
        b = djb.tabular_anisotropic(src, 90, 90, True, ctx=ctx)
    elif name == "tabular_abc_sample":
        src = djb.abc("gold-metallic-paint", ctx=ctx)
        b = djb.tabular(src, 90, True, ctx=ctx)
    elif name.startswith("tabular"):
        src = djb.ggx(ctx=ctx)
        b = djb.tabular(src, 90, True, ctx=ctx)
    elif name == "ggx_evalp_is":
        src, b = None, djb.ggx(ctx=ctx)
    elif name == "beckmann_evalp_is":
        src, b = None, djb.beckmann(ctx=ctx)
    elif name == "lean_evalp_pdf":
        src, b = None, djb.beckmann(djb.fresnel.schlick((1.0, 0.71, 0.29)), True, ctx=ctx)
    else:
        src, b = None, djb.abc("gold-metallic-paint", ctx=ctx)
    out = torch.empty((3, n), dtype=torch.float32, device=dev)
    vout = djb._Vec(out)
    if name.endswith("_sample"):           # sample() with on-chip uniforms: o -> i
        def step():
            chk(lib.djb_sample_rng_batch(ctx._h, b._h, C.c_int64(n), C.c_uint32(synth.SEED_U1), C.c_uint32(synth.SEED_U2), C.c_uint64(0),
                                         C.byref(vo.view), None, C.byref(vout.view)))
        return step, (o, out, b, src, vo, vout)
    if name.endswith("_evalp_is"):
        u1 = djb.gen_uniforms(n, synth.SEED_U1, ctx=ctx); u2 = djb.gen_uniforms(n, synth.SEED_U2, ctx=ctx)
        wi = torch.empty((3, n), dtype=torch.float32, device=dev); pdf = torch.empty((n,), dtype=torch.float32, device=dev)
        vwi = djb._Vec(wi)
        p = djb.microfacet.params.elliptic(0.2, 0.5, 0.7)
        def step():
            chk(lib.djb_evalp_is_batch(ctx._h, b._h, C.c_int64(n), C.c_void_p(u1.data_ptr()), C.c_void_p(u2.data_ptr()), C.byref(vo.view),
                                       C.byref(p._p), C.byref(vout.view), C.byref(vwi.view), C.c_void_p(pdf.data_ptr()), C.c_int(0)))
        return step, (o, out, b, p, u1, u2, wi, pdf, vo, vout, vwi)
    i = djb.gen_directions(n, synth.SEED_I, ctx=ctx)
    vi = djb._Vec(i)
    if name == "abc_evalp":
        def step():
            chk(lib.djb_evalp_batch(ctx._h, b._h, C.c_int64(n), C.byref(vi.view), C.byref(vo.view), None, C.byref(vout.view), C.c_int(0)))
        return step, (i, o, out, b, vi, vo, vout)
    pdf = torch.empty((n,), dtype=torch.float32, device=dev)
    if name == "lean_evalp_pdf":
        g = torch.Generator(device=dev); g.manual_seed(7)
        lean = torch.empty((n, 5), dtype=torch.float32, device=dev)
        lean[:, 0:2] = (torch.rand((n, 2), generator=g, device=dev) - 0.5) * 0.2          # E1, E2: mean slopes
        lean[:, 2:4] = torch.rand((n, 2), generator=g, device=dev) * 0.05 + 0.01         # E3, E4: second moments
        lean[:, 4] = (torch.rand((n,), generator=g, device=dev) - 0.5) * 0.01            # E5
        base = djb.microfacet.params.isotropic(0.1)
        def step():
            chk(lib.djb_eval_lean_batch(ctx._h, b._h, C.c_int64(n), C.byref(vi.view), C.byref(vo.view), C.byref(base._p), C.c_float(1.0), C.c_int(0),
                                        C.c_void_p(lean.data_ptr()), C.c_int(6), C.byref(vout.view), C.c_void_p(pdf.data_ptr()), C.c_void_p(0), C.c_int(0)))
        return step, (i, o, out, pdf, b, base, lean, vi, vo, vout)
    def step():             # tabular / tabular_anisotropic eval + pdf fused
        chk(lib.djb_eval_pdf_batch(ctx._h, b._h, C.c_int64(n), C.byref(vi.view), C.byref(vo.view), None, C.c_int(0), C.byref(vout.view),
                                   C.c_void_p(pdf.data_ptr()), C.c_int(0)))
    return step, (i, o, out, pdf, b, src, vi, vo, vout)


def one_pair_calls(with_reference):
    """What ONE call of the djb:: surface costs (the reference's real callers make one-pair virtual calls from render threads):
    examples/scalar_latency on a GPU object (answered by its host twin, no launch) and, as the reported CPU baseline, the same source
    built on the real reference (oracle/_ref/scalar_latency) on the same host cores."""
    import subprocess

    def run(exe, env_extra=None):
        if not os.path.exists(exe):
            return None
        env = dict(os.environ, DJB_QUIET="1")
        env.update(env_extra or {})
        try:
            r = subprocess.run([exe, "16"], capture_output=True, text=True, timeout=120, env=env)
        except Exception:
            return None
        out = {}
        for line in r.stdout.splitlines():
            if "ns per call" in line:
                out[" ".join(line.split()[:-4])] = float(line.split()[-4])
            elif "threads on one ggx object" in line:
                out["16_threads_one_object_M_calls_per_s"] = float(line.split(":")[1].split("M calls/s")[0])
        return out or None

    rec = {"unit": "ns per call", "gpu_object_host_twin": run(os.path.join(ROOT, "examples", "scalar_latency")),
           "cpu_context": run(os.path.join(ROOT, "examples", "scalar_latency"), {"DJB_DEVICE": "cpu"}),
           "what": "one (i, o) pair per call through the C++ facade's virtuals, single thread; `brdf*->eval, dependent` = each call's input "
                   "depends on the previous result (the latency of one call); batches are the GPU's business, these are not"}
    if with_reference:
        ref = run(os.path.join(ROOT, "oracle", "_ref", "scalar_latency"))
        if ref:
            rec["cpu_baseline"] = {"kind": "reference", "cores": 1, "sample": "400k calls per operator, the same source on the reference's header (-O2)", **ref}
    return rec


def cpu_baseline(name, synth, budget_s=12.0):
    """The CPU path on this host's cores, bounded sample.  Prefers the real reference (oracle/_ref)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oraclelib
    ref_path = os.path.join(ROOT, "oracle", "_ref", "libdjb_ref.so")
    kind = "reference" if os.path.exists(ref_path) else "port"
    L = oraclelib.CheckerLib(ref_path, "ref_") if kind == "reference" else oraclelib.oracle()
    cores = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    par = None
    if name.startswith("merl_eval"):
        if kind == "reference":
            path = "/tmp/djb_bench_cpu.binary"
            synth.write_merl_binary(path, synth.merl_table(0.3))
            b = L.merl(path)
        else:
            b = L.merl_from_table(synth.merl_table(0.3))
        op = "eval"
    elif name in ("ggx_eval_pdf", "ggx_eval_pdf_contract", "ggx_unpolarized_eval_pdf", "ggx_unpolarized_eval_pdf_contract"):
        fres = ("unpolarized", 1.5, 1.8, 2.4) if "unpolarized" in name else ("ideal",) if GGX_FRESNEL == "ideal" else ("schlick", 1.0, 0.71, 0.29)
        b, op, par = L.microfacet("ggx", fres, True), "eval", ("elliptic", GGX_ALPHA, GGX_ALPHA, 0.0)
    elif name in ("sgd_eval", "sgd_eval_contract"):
        b, op = L.sgd("gold-metallic-paint"), "eval"
    elif name in ("beckmann_sample", "beckmann_sample_contract"):
        b, op, par = L.microfacet("beckmann", ("ideal",), True), "sample", ("elliptic", 0.2, 0.5, 0.7)
    elif name in ("utia_eval", "utia_eval_contract"):
        path = "/tmp/djb_bench_cpu_utia.bin"
        np.random.default_rng(11).uniform(0.0, 120.0, size=3 * 288 * 288).tofile(path)
        b, op = L.utia(path), "eval"
    elif name == "merl_fit_files":   # the reference's own driver (examples/merl_params.cpp) on a few of the files
        import subprocess
        exe = os.path.join(ROOT, "oracle", "_ref", "merl_params")
        paths = synth_merl_files(100, synth)[:4]
        if os.path.exists(exe):
            t0 = time.perf_counter()
            subprocess.run([exe] + paths, cwd="/tmp", check=True, stdout=subprocess.DEVNULL)
            dt = time.perf_counter() - t0
            return {"value": len(paths) / dt, "unit": "materials/s", "cores": 1, "kind": "reference",
                    "sample": f"examples/merl_params (reference, -O2) on {len(paths)} of the same files, 1 thread, "
                              f"page cache warm: {dt / len(paths):.3f} s per file"}
        t0 = time.perf_counter()
        for p in paths:
            L.tabular(L.merl(p), 90, True)
        dt = time.perf_counter() - t0
        return {"value": len(paths) / dt, "unit": "materials/s", "cores": 1, "kind": kind,
                "sample": f"load + tabular(merl, 90) on {len(paths)} of the same files, 1 thread"}
    else:   # merl_fit: materials / s, one fit per thread
        tab = synth.merl_table(*synth.material_recipe(0))
        if kind == "reference":
            path = "/tmp/djb_bench_cpu.binary"
            synth.write_merl_binary(path, tab)
            src = L.merl(path)
        else:
            src = L.merl_from_table(tab)
        t0 = time.perf_counter(); L.tabular(src, 90, True); t1 = time.perf_counter() - t0
        reps = max(1, min(8, int(budget_s / max(t1, 1e-3) / 2)))
        ths = [threading.Thread(target=lambda: [L.tabular(src, 90, True) for _ in range(reps)]) for _ in range(cores)]
        t0 = time.perf_counter()
        for t in ths: t.start()
        for t in ths: t.join()
        dt = time.perf_counter() - t0
        return {"value": cores * reps / dt, "unit": "materials/s", "cores": cores, "kind": kind,
                "sample": f"{cores * reps} fits of one synthetic MERL table (res 90), {cores} threads; "
                          f"single-thread {1.0 / t1:.2f} materials/s", "single_thread": 1.0 / t1}

    def run(n, threads):
        i = synth.directions_aos(n, synth.SEED_I); o = synth.directions_aos(n, synth.SEED_O)
        u1 = synth.uniforms(n, synth.SEED_U1); u2 = synth.uniforms(n, synth.SEED_U2)
        bounds = np.linspace(0, n, threads + 1).astype(np.int64)
        chunks = [slice(int(a), int(b)) for a, b in zip(bounds[:-1], bounds[1:])]   # contiguous views

        def work(idx):
            if op == "sample":
                L.sample(b, u1[idx], u2[idx], o[idx], par)
            else:
                L.eval(b, i[idx], o[idx], par, "eval")
                if name.startswith("ggx_"):
                    L.eval(b, i[idx], o[idx], par, "pdf")
        ths = [threading.Thread(target=work, args=(c,)) for c in chunks]
        t0 = time.perf_counter()
        for t in ths: t.start()
        for t in ths: t.join()
        return n / (time.perf_counter() - t0)

    r1 = run(500_000, 1)                                       # single thread: the reference's own design
    # thread-scaling probe (same per-thread work): the box may expose more logical CPUs than it lets
    # one process use, so report the best aggregate rate and the thread count that achieved it
    best, best_t, probe = r1, 1, {1: r1}
    t0 = time.perf_counter()
    for t in sorted({4, 16, 64, cores // 2, cores} - {0, 1}):
        if t > cores or time.perf_counter() - t0 > budget_s:
            continue
        probe[t] = run(int(min(600_000 * t, 1.2e8)), t)
        if probe[t] > best:
            best, best_t = probe[t], t
    quota = cpu_quota()
    # cores: the CPUs that actually did the work -- the thread count of the best run, capped by the container's CPU quota (128
    # threads under a 16-CPU quota get 16 CPUs' worth of time); the thread count itself is kept as `threads`
    eff = best_t if quota is None else max(1, min(best_t, int(round(quota))))
    return {"value": best, "unit": "evals/s" if op != "sample" else "samples/s", "cores": eff, "threads": best_t, "kind": kind,
            "sample": f"600k units per thread of the same synthetic workload, thread counts {sorted(probe)} of "
                      f"{cores} usable logical CPUs (ctypes releases the GIL; the reference object is const); "
                      f"best at {best_t} threads; single-thread {r1:.3e}/s"
                      + (f"; the container's CPU quota (cgroup cpu.max) is {quota:g} CPUs: the rate stops scaling there, "
                         f"whatever the thread count" if quota is not None else ""),
            "cpu_quota_cpus": quota, "single_thread": r1, "scaling_probe": {str(k): v for k, v in probe.items()}}


def cpu_quota():
    """CPUs' worth of time the container may use (cgroup v2 cpu.max / v1 cfs quota), or None when unlimited or unknown"""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except Exception:
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / per
    except Exception:
        return None


def rank_placement(torch, dist, rank, local, world, pin):
    """Where this rank runs: device id, PCI bus id, the GPU's NUMA node and the CPUs the rank's host threads (the file
    pipeline's reader pool above all) may use.  With more than one rank per node (pin=True) each rank is confined to its
    share of the CPUs of its GPU's NUMA node -- ranks whose GPUs hang off the same node split that node's CPUs between
    them -- and DJB_READER_THREADS defaults to that share (at most 32: profiles/r03/fit_files_rates.txt), so that eight
    ranks do not each start a 32-thread pool on the same cores.  Returns the list of every rank's record (rank order)."""
    info = {"rank": rank, "local_rank": local, "device": local, "pci_bus_id": None, "numa_node": None}
    try:
        pr = torch.cuda.get_device_properties(local)
        info["name"] = pr.name
        info["pci_bus_id"] = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
        with open(f"/sys/bus/pci/devices/{info['pci_bus_id']}/numa_node") as f:
            info["numa_node"] = int(f.read().strip())
    except Exception:
        pass
    allowed = sorted(os.sched_getaffinity(0))
    node_cpus = allowed
    if info["numa_node"] is not None and info["numa_node"] >= 0:
        try:
            cl = open(f"/sys/devices/system/node/node{info['numa_node']}/cpulist").read().strip()
            cpus = set()
            for part in cl.split(","):
                a, _, b = part.partition("-")
                cpus.update(range(int(a), int(b or a) + 1))
            node_cpus = [c for c in allowed if c in cpus] or allowed
        except Exception:
            pass
    everyone = [info]
    if world > 1:
        everyone = [None] * world
        dist.all_gather_object(everyone, info)
    if pin and world > 1:
        peers = [r["rank"] for r in everyone if r["numa_node"] == info["numa_node"]]      # ranks sharing this NUMA node's CPUs
        k, j = len(peers), peers.index(rank)
        mine = node_cpus[j * len(node_cpus) // k:(j + 1) * len(node_cpus) // k] or node_cpus
        try:
            os.sched_setaffinity(0, mine)
        except OSError:
            mine = allowed
        os.environ.setdefault("DJB_READER_THREADS", str(max(2, min(32, len(mine)))))
        info["cpus"] = len(mine)
        info["cpu_range"] = f"{mine[0]}-{mine[-1]}"
        info["reader_threads"] = int(os.environ["DJB_READER_THREADS"])
        everyone = [None] * world
        dist.all_gather_object(everyone, info)
    else:
        info["cpus"] = len(allowed)
    return everyone


# What the 1 / 2 / 4 / 8-GPU curve of the fit legs should look like, so that the first real SCALE record
# can be held against a stated expectation.  End to end (files -> alphas): a fixed part (slot plan look-up, waking the reader
# pool, the 97 KB-per-material upload, one k_fit launch of ~0.35 ms, the host-side barrier) plus the gather of 5 545 x 3
# doubles per file, which is the only part that shrinks with the rank's share of the files; compute only: one wave of
# workgroups at every N (100 materials or 13: the launch floor); dense upload: each rank's PCIe link.
def scaling_model_ms(world):
    ns = sorted({1, 2, 4, 8, world})
    return {"merl_fit_files_100.wall_ms": {str(n): round(1.15 + 1.4 / n, 2) for n in ns},
            "merl_fit_100.wall_ms": {str(n): 0.34 if n == 1 else 0.30 for n in ns},
            "merl_fit_files_100.dense_upload.wall_ms": {str(n): round(2.0 + 134.0 / n, 1) for n in ns},
            "primary.value": "N x the single-GPU rate (weak scaling, no data-path collective)",
            "source": "model: N=1 terms measured on one MI355X (profiles/r04), 1/N applied to the per-file gather only; NOT a measurement",
            # BASELINE configs[4] on this library: 100 files -> alphas is 2.3-2.5 ms on ONE GPU, of which only the ~1.4 ms gather of
            # 5 545 x 3 doubles per file divides by N; the fit is one wave of workgroups at every N.  A second GPU buys ~0.7 ms, eight
            # ~1.2 ms -- less than creating their contexts costs.  The job's 1 / 2 / 4 / 8 curve is reported because BASELINE asks for
            # it, not because it is the way to run this job.
            "n_gpus_useful_for_config5": 1,
            "n_gpus_useful_reason": "files -> alphas for 100 materials takes ~2.4 ms on one GPU (sparse gather 1.4-1.9 ms + one 0.34 ms fit "
                                    "launch + fixed 0.3 ms); only the gather shrinks with N (model: 2.55 ms at N=1 -> 1.32 ms at N=8), and one "
                                    "djb_ctx_create costs more than that; multi-GPU pays for the dense-upload form (184 ms -> 2 + 134/N ms) "
                                    "and for the weak-scaling eval workloads, not for this fit"}


def static_profile(name, n, launch_ms):
    """What the tracked profile summaries say about this leg's kernels (profiles/pmc_<name>.json: separate rocprofv3 --pmc passes of
    tools/profile_bench.sh; profiles/valu_<name>.json: SQ_INSTS_VALU_* passes of tools/instmix.sh), scaled to this run's batch -- attached
    only if they list exactly the kernels the leg launches (LAUNCHES); static, NOT measured in this run."""
    out = {}
    try:
        pj = json.load(open(os.path.join(ROOT, "profiles", f"pmc_{name}.json")))
        ok, listed = profile_matches(name, pj.get("kernels", []))
        if ok and pj.get("hbm_bytes_per_launch") and pj.get("units_per_launch"):
            per_unit = pj["hbm_bytes_per_launch"] / float(pj["units_per_launch"])
            out["traffic_bytes_per_unit"] = per_unit
            ob = WORKLOADS[name][1]
            if ob:
                out["traffic_over_algorithmic"] = per_unit / ob
            out["traffic_source"] = "static: profiles/pmc_%s.json (%s, tree %s)" % (name, pj.get("round", "?"), pj.get("tree", "?"))
        elif not ok:
            out["traffic_source"] = "STALE: profiles/pmc_%s.json lists %s, the leg launches %s" % (name, listed, sorted(LAUNCHES.get(name, [])))
    except Exception:
        pass
    try:
        v = json.load(open(os.path.join(ROOT, "profiles", f"valu_{name}.json")))
        ok, listed = profile_matches(name, [k.get("kernel", "") for k in v.get("kernels", [])])
        if ok:
            issue_ms = v["slots_per_unit"] * n / 64.0 * 1.155e-6 / 1024.0
            out["valu"] = {"insts_per_unit": v["insts_per_unit"], "slots_per_unit": v["slots_per_unit"], "frac_of_issue": issue_ms / launch_ms,
                           "source": "static: profiles/valu_%s.json (%s, tree %s)" % (name, v.get("round", "?"), v.get("tree", "?"))}
    except Exception:
        pass
    return out


def merl_order_lever(djb, synth, ctx, torch, local, n=1 << 28, tile=4096):
    """The one lever the MERL look-up leaves to its caller: the ORDER of the batch.  The same 2^28 random pairs (the headline
    distribution) evaluated (a) as generated, (b) with every consecutive tile of 4096 pairs ordered by bin key -- the most a per-workgroup
    LDS bucketing pass inside the kernel could achieve, here done outside the timed region, so the row is an upper bound for it --
    (c) sorted by bin key over the whole batch (what a wavefront renderer that already sorts its hits gets by appending the key);
    plus the cost of producing the keys (djb_merl_bin_keys_batch: tier-1 arithmetic only, 28 B per pair)."""
    lib, C = djb._lib.load(), ctypes
    dev = f"cuda:{ctx.device}"
    i = djb.gen_directions(n, synth.SEED_I, ctx=ctx); o = djb.gen_directions(n, synth.SEED_O, ctx=ctx)
    m = djb.merl.from_table(synth.merl_table(0.3), ctx=ctx)
    out = torch.empty((3, n), dtype=torch.float32, device=dev)
    keys = torch.empty((n,), dtype=torch.int32, device=dev)
    vout = djb._Vec(out)

    def evaluator(a, b):
        va, vb = djb._Vec(a), djb._Vec(b)

        def st():
            djb._lib.check(lib.djb_eval_batch(ctx._h, m._h, C.c_int64(n), C.byref(va.view), C.byref(vb.view), None, C.byref(vout.view), C.c_int(0)))
        st.keep = (va, vb, a, b)
        return st
    vi, vo = djb._Vec(i), djb._Vec(o)

    def key_step():
        djb._lib.check(lib.djb_merl_bin_keys_batch(ctx._h, C.c_int64(n), C.byref(vi.view), C.byref(vo.view), C.c_void_p(keys.data_ptr()), C.c_int(0)))
    res = {"pairs": n, "tile": tile, "algorithmic_bytes_per_unit": 36}

    def row(ms, spread, bytes_per_unit=36):
        return {"ms_per_step": ms, "ms_min_median_max": spread, "value": n / (ms * 1e-3), "unit": "evals/s",
                "roofline_frac": n * bytes_per_unit / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
    ms, spread, _ = timed_leg(key_step, torch, local)
    res["key_kernel"] = dict(row(ms, spread, 28), unit="keys/s", algorithmic_bytes_per_unit=28, kernel="k_merl_keys")
    ms, spread, _ = timed_leg(evaluator(i, o), torch, local)
    res["as_generated"] = row(ms, spread)
    # (b) order inside tiles of `tile` pairs
    kt = keys.view(n // tile, tile)
    perm = torch.sort(kt, dim=1).indices + (torch.arange(n // tile, device=dev, dtype=torch.int64) * tile).unsqueeze(1)
    perm = perm.reshape(-1)
    ib = torch.stack([i[c][perm] for c in range(3)]).contiguous(); ob = torch.stack([o[c][perm] for c in range(3)]).contiguous()
    ms, spread, _ = timed_leg(evaluator(ib, ob), torch, local)
    res["tile_bucketed"] = row(ms, spread)
    del kt, ib, ob
    # (c) the whole batch sorted by key
    perm = torch.sort(keys.to(torch.int64)).indices
    ig = torch.stack([i[c][perm] for c in range(3)]).contiguous(); og = torch.stack([o[c][perm] for c in range(3)]).contiguous()
    ms, spread, _ = timed_leg(evaluator(ig, og), torch, local)
    res["sorted_by_key"] = row(ms, spread)
    del perm, ig, og
    torch.cuda.empty_cache()
    a, b, c = res["as_generated"]["ms_per_step"], res["tile_bucketed"]["ms_per_step"], res["sorted_by_key"]["ms_per_step"]
    res["what"] = ("the same 2^28 random pairs in three orders; tile_bucketed is an UPPER BOUND for an in-kernel per-workgroup LDS bucketing pass "
                   "(the ordering was done outside the timed region); a tile of %d pairs touches ~%d distinct table lines of 136 688 either way, so "
                   "ordering inside it changes neither the lines a launch touches per unit time nor the L2 hit rate" % (tile, tile))
    res["break_even"] = ("sorting buys %.2f ms per 2^28 pairs (%.2f -> %.2f) and costs the key kernel (%.2f ms) plus a sort of 2^28 (key, index) "
                         "records and a 24 B/pair gather -- several times the look-up itself on this chip -- so it pays only where the caller "
                         "sorts anyway (wavefront renderers sorting hits by material: append the 21 key bits)" % (a - c, a, c, res["key_kernel"]["ms_per_step"]))
    return res


def gpu_state(device=0):
    """Shader clock / power / temperature of the GPU as the kernel driver reports them right now (sysfs hwmon of the card; no subprocess,
    ~0.1 ms).  Called while a leg's launches are in flight, so the figures are those of the loaded chip.  None where a file is absent."""
    import glob
    out = {}
    cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device"))
    cards = [c for c in cards if os.path.exists(os.path.join(c, "pp_dpm_sclk")) or glob.glob(os.path.join(c, "hwmon", "hwmon*", "freq1_input"))]
    if not cards:
        return None
    c = cards[min(device, len(cards) - 1)]

    def rd(path):
        try:
            return open(path).read().strip()
        except Exception:
            return None
    for hw in glob.glob(os.path.join(c, "hwmon", "hwmon*")):
        v = rd(os.path.join(hw, "freq1_input"))
        if v and v.isdigit():
            out["sclk_mhz"] = int(v) / 1e6
        v = rd(os.path.join(hw, "freq2_input"))
        if v and v.isdigit():
            out["mclk_mhz"] = int(v) / 1e6
        v = rd(os.path.join(hw, "power1_average")) or rd(os.path.join(hw, "power1_input"))
        if v and v.isdigit():
            out["power_w"] = int(v) / 1e6
        v = rd(os.path.join(hw, "temp1_input"))
        if v and v.lstrip("-").isdigit():
            out["temp_c"] = int(v) / 1e3
    if "sclk_mhz" not in out:
        v = rd(os.path.join(c, "pp_dpm_sclk"))
        for line in (v or "").splitlines():
            if line.rstrip().endswith("*"):
                try:
                    out["sclk_mhz"] = float(line.split(":")[1].lower().replace("mhz", "").replace("*", "").strip())
                except Exception:
                    pass
    return out or None


def timed_leg(st, torch, device, reps=10, warm=10):
    """warm untimed launches (steady clocks: the launches are 1-20 ms and the GPU idled during the CPU legs), then reps launches with an
    event between each on the stream the library launches on (torch's current stream = the context's), so that every launch has its own
    duration; the GPU's clock / power are read while they are in flight.  -> (mean ms, {min, median, max}, gpu state mid-leg)"""
    for _ in range(warm):
        st()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for k in range(reps):
        st()
        ev[k + 1].record()
    state = gpu_state(device)                 # the launches are asynchronous: the chip is busy with them now
    torch.cuda.synchronize()
    ms = sorted(ev[k].elapsed_time(ev[k + 1]) for k in range(reps))
    return sum(ms) / reps, {"min": ms[0], "median": ms[reps // 2], "max": ms[-1]}, state


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    from dj_brdf_amd import djb, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        sys.exit(f"bench.py --gpus {args.gpus} must be launched with torch.distributed.run --nproc-per-node {args.gpus}")
    assert torch.cuda.is_available() and djb.device_count() > 0, "bench.py needs MI355X GPUs; there is no CPU path"
    # DJB_BENCH_SHARE_GPU=1 (harness self-test on a one-GPU box only): every rank uses device 0 and the control-plane
    # collectives (barrier, MAX of the timings) run over gloo -- RCCL refuses two ranks on one device.  The data path has
    # no collective either way.  Numbers from such a run say nothing about scaling and are labelled.
    share_gpu = os.environ.get("DJB_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local = 0
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local}"))
    red_dev = "cpu" if share_gpu else f"cuda:{local}"      # where the control-plane reductions live
    ranks = rank_placement(torch, dist, rank, local, world, pin=world > 1 and os.environ.get("DJB_BENCH_NO_PIN") != "1")
    ctx = djb.Context(local)    # runs on torch's current stream of this device

    global GGX_ALPHA, GGX_FRESNEL
    GGX_ALPHA, GGX_FRESNEL = args.alpha, args.fresnel
    name = args.workload
    n_default, bytes_per_unit, unit, kernel = WORKLOADS[name]
    n = args.n or args.selftest_n or n_default
    step, keep = make_step(name, n, djb, synth, ctx, torch)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    ctx.timer_start()                       # HIP event on the stream the kernels are launched on
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]      # the same stream: one mark per step boundary
    marks[0].record()
    for k in range(args.steps):
        step()
        marks[k + 1].record()
    state_in_flight = gpu_state(local)      # the launches are asynchronous: read while the chip is busy with them
    ev_ms = ctx.timer_stop_ms()             # records + synchronises the closing event
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt, ev_ms], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, ev_ms = float(t[0]), float(t[1])

    # BASELINE.json's second figure: wall time of the 100-material MERL fit on N GPUs (strong scaling:
    # material m -> rank m mod N, no exchange; tables resident in HBM).  Every rank takes part.
    fit100 = fitfiles = None
    want_secondary = not args.no_secondary and name == "merl_eval" and args.n is None
    finish(step)
    if want_secondary:
        del step, keep
        torch.cuda.empty_cache()
        st, kp = make_step("merl_fit", list(range(rank, 100, world)), djb, synth, ctx, torch)
        barrier()
        ctx.timer_start()
        st()                                # the first call of this shape on the context also builds the material-independent tables
        fit_first_ms = ctx.timer_stop_ms()
        barrier()
        ctx.timer_start()
        for _ in range(3):
            st()
        fit_ms = ctx.timer_stop_ms() / 3
        barrier()
        if world > 1:
            t = torch.tensor([fit_ms, fit_first_ms], dtype=torch.float64, device=red_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            fit_ms, fit_first_ms = float(t[0]), float(t[1])
        fit100 = {"materials": 100, "n_gpus": world, "wall_ms": fit_ms, "first_call_wall_ms": fit_first_ms, "scaling": "strong",
                  "value": 100 / (fit_ms * 1e-3), "unit": "materials/s",
                  "what": "compute only: tables resident in HBM, one k_fit launch per rank; wall_ms is a call on a context that has fitted at this "
                          "resolution before (it keeps the fit's material-independent geometry tables -- directions, K-matrix integrals, MERL bins "
                          "of the query slots -- which the reference recomputes per material); first_call_wall_ms is the call that builds them"}
        del st, kp
        torch.cuda.empty_cache()
        # ... and end to end, files -> alphas (the reference driver's loop is file to file): rank r takes files r, r+N, ...
        from dj_brdf_amd import merl_params
        # every rank writes the files it will read (its own directory under /tmp), before the timed region
        all_mine = synth_merl_files(100, synth, rank=rank, only=list(range(rank, 100, world)))
        def files_leg(dense):
            djb.set_fit_files_dense(ctx, dense)
            try:
                merl_params.fit_files_on(ctx, all_mine[:2])        # page cache of the first files, kernels loaded
                barrier()
                # first call of this shape on the context: also starts the reader threads and pins the staging buffer
                t_files = time.perf_counter()
                merl_params.fit_files_on(ctx, all_mine)
                barrier()
                first = time.perf_counter() - t_files
                time.sleep(0.1)                                    # the previous call's mappings are released off the critical path
                barrier()
                t_files = time.perf_counter()
                _, _, tim = merl_params.fit_files_on(ctx, all_mine)   # the same job again: every file gathered, uploaded and fitted again
                barrier()
                wall = time.perf_counter() - t_files
                # ... and once with the files' page cache dropped first: what 100 files cost the first time they are read
                time.sleep(0.1)
                still_resident = drop_page_cache(all_mine)
                barrier()
                t_files = time.perf_counter()
                merl_params.fit_files_on(ctx, all_mine)
                barrier()
                cold = time.perf_counter() - t_files
            finally:
                djb.set_fit_files_dense(ctx, False)
            tt = torch.tensor([wall, tim["total_s"], tim["load_s"], tim["fit_s"], first, cold], dtype=torch.float64, device=red_dev)
            if world > 1:
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            wall, f_total, f_load, f_fit, first, cold = (float(x) for x in tt)
            return {"wall_ms": wall * 1e3, "first_call_wall_ms": first * 1e3, "value": 100 / wall, "unit": "materials/s",
                    "cold_cache_wall_ms": cold * 1e3,
                    "cold_cache_note": ("page cache of this rank's files dropped before the call (fsync + posix_fadvise(DONTNEED)); resident "
                                        "fraction afterwards by mincore: %s%s" % ("n/a" if still_resident is None else "%.3f" % still_resident,
                                        " -- the files sit on a file system that keeps its pages (tmpfs): this is NOT a cold read"
                                        if still_resident is not None and still_resident > 0.5 else "")),
                    "pipeline_ms": {"total": f_total * 1e3, "load": f_load * 1e3, "fit": f_fit * 1e3},
                    "bytes_read_this_rank": tim["bytes"]}
        sparse, dense = files_leg(False), files_leg(True)
        fitdir = None
        if args.merl_dir:          # the same leg on real MERL files (rank r takes files r, r + N, ...), warm then cold
            import glob
            real = sorted(glob.glob(os.path.join(args.merl_dir, "*.binary")))
            mine_real = real[rank::world]
            if mine_real:
                merl_params.fit_files_on(ctx, mine_real[:1])
                barrier(); t0f = time.perf_counter()
                ab_r, ag_r, _ = merl_params.fit_files_on(ctx, mine_real)
                barrier(); warm_r = time.perf_counter() - t0f
                res_r = drop_page_cache(mine_real)
                barrier(); t0f = time.perf_counter()
                merl_params.fit_files_on(ctx, mine_real)
                barrier(); cold_r = time.perf_counter() - t0f
                fitdir = {"dir": args.merl_dir, "files": len(real), "files_this_rank": len(mine_real), "wall_ms": warm_r * 1e3,
                          "cold_cache_wall_ms": cold_r * 1e3, "resident_after_drop": res_r,
                          "first_materials": [[os.path.basename(pth).split(".")[0], "%.3f" % a, "%.3f" % g] for pth, a, g in zip(mine_real[:5], ab_r[:5], ag_r[:5])]}
            else:
                fitdir = {"dir": args.merl_dir, "files": 0, "note": "no *.binary files found"}
        fitfiles = {"materials": 100, "n_gpus": world, "scaling": "strong", **sparse,
                    "what": "end to end, files -> alphas, max over ranks: table indices a tabular(merl, 90) fit reads computed on the GPU, "
                            "worker threads gather those 5 545 x 3 doubles per file from the mapped files (page cache warm), "
                            "97 KB per material -> HBM, one k_fit launch per rank; wall_ms is the second call of this shape on the context "
                            "(reader threads parked, staging buffer pinned, slot plan cached), first_call_wall_ms the one that set those up",
                    "dense_upload": {**dense, "what": "same job with every 35 MB table uploaded and converted in full (pread -> 4 MiB pinned "
                                                      "chunk ring -> H2D -> k_merl_convert), as in round 1: same alphas"}}

    if rank == 0:
        value = world * n * args.steps / dt
        launch_ms = ev_ms / args.steps
        if bytes_per_unit is not None:
            achieved = n * bytes_per_unit / (launch_ms * 1e-3) / 1e9
            roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": achieved / HBM_PEAK_GBS, "traffic": None, "kernel": kernel,
                        "algorithmic_bytes_per_unit": bytes_per_unit, "launch_ms": launch_ms}
            pmc = os.path.join(ROOT, "profiles", f"pmc_{name}.json")
            if os.path.exists(pmc):      # HBM bytes per launch from rocprofv3 PMC passes (profiles/README.md)
                try:
                    pj = json.load(open(pmc))
                    ok, listed = profile_matches(name, pj.get("kernels", []))
                    if ok:
                        roofline["traffic"] = pj.get("hbm_bytes_per_launch")
                        roofline["traffic_source"] = ("static: read from profiles/pmc_%s.json (separate rocprofv3 --pmc passes of "
                                                      "tools/profile_bench.sh, %s, tree %s; kernels %s = the ones this run launched), NOT measured in this run"
                                                      % (name, pj.get("round", "round 1"), pj.get("tree", "not recorded"), listed))
                        roofline["traffic_note"] = pj.get("note")
                        if pj.get("gather_miss_bytes_per_launch") is not None:      # the table gathers' share (line fills out of the Infinity Cache)
                            roofline["traffic_gather_miss_bytes"] = pj["gather_miss_bytes_per_launch"]
                        # the same counters as memory-side REQUESTS: a read request moves 128 bytes (FETCH_SIZE tallies it at 64), the L2 writes
                        # back in 64-byte requests; every memory-bound kernel of this library sustains 60-63 G requests/s (= 8 TB/s in 128-byte
                        # reads; profiles/r04/fabric_request_rate.txt), so requests / 62e9 is the time the memory system needs for this
                        # access pattern -- for the MERL launch the line fills of the gather misses are 45 % of the requests
                        if pj.get("FETCH_SIZE_KB_per_launch") and pj.get("WRITE_SIZE_KB_per_launch") and pj.get("units_per_launch"):
                            rd = pj["FETCH_SIZE_KB_per_launch"] * 1024.0 / 64.0
                            wr = pj["WRITE_SIZE_KB_per_launch"] * 1024.0 / 64.0
                            upl = float(pj["units_per_launch"])
                            floor_ms = (rd + wr) / 62e9 * 1e3 * (n / upl)
                            roofline["request_model"] = {
                                "read_requests_per_unit": rd / upl, "write_requests_per_unit": wr / upl, "sustained_G_requests_per_s": 62.0,
                                "memory_system_floor_ms": floor_ms, "launch_ms_over_floor": launch_ms / floor_ms,
                                "note": "static (same PMC passes as `traffic`): 128-byte read requests + 64-byte write requests per unit, and the time "
                                        "they take at the 62 G requests/s this memory system sustains; `frac` above prices the ALGORITHMIC bytes "
                                        "against 8 TB/s as the contract asks, this object says how far the launch is from what its access "
                                        "pattern allows"}
                    else:
                        roofline["traffic_source"] = ("STALE: profiles/pmc_%s.json lists kernels %s, this workload launches %s -- not attached"
                                                      % (name, listed, sorted(LAUNCHES.get(name, []))))
                except Exception:
                    pass
            vj = os.path.join(ROOT, "profiles", f"valu_{name}.json")
            if os.path.exists(vj):       # VALU issue load beside the HBM fraction: tools/instmix.sh + tools/valu_report.py
                try:
                    v = json.load(open(vj))
                    ok, listed = profile_matches(name, [k.get("kernel", "") for k in v.get("kernels", [])])
                    issue_ms = v["slots_per_unit"] * n / 64.0 * 1.155e-6 / 1024.0
                    if ok:
                        roofline["valu"] = {"insts_per_unit": v["insts_per_unit"], "slots_per_unit": v["slots_per_unit"],
                                            "frac_of_issue": issue_ms / launch_ms, "kernels": listed,
                                            "note": "issue-bound" if issue_ms / launch_ms > 0.9 else None,
                                            "source": "static: instruction mix from profiles/valu_%s.json (SQ_INSTS_VALU_* passes of rocprofv3, %s, tree %s; "
                                                      "issue slots per class from tools/valu_cost_probe.hip, one slot = 1.155 ns per wave-instruction "
                                                      "per SIMD, 1024 SIMDs), divided by this run's launch_ms" % (name, v.get("round", "?"), v.get("tree", "not recorded"))}
                    else:
                        roofline["valu"] = {"stale": True, "note": "profiles/valu_%s.json lists kernels %s, this workload launches %s -- not attached"
                                                                   % (name, listed, sorted(LAUNCHES.get(name, [])))}
                except Exception:
                    pass
        else:
            # the fit moves ~5.5k table reads per material: HBM traffic is negligible, the kernel is
            # latency/VALU-bound; report the achieved rate only
            roofline = {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None,
                        "traffic": None, "kernel": kernel, "launch_ms": launch_ms}
        per_step = sorted(marks[k].elapsed_time(marks[k + 1]) for k in range(args.steps))
        roofline["launch_ms_min_median_max"] = {"min": per_step[0], "median": per_step[len(per_step) // 2], "max": per_step[-1]}
        roofline["gpu_in_flight"] = state_in_flight       # shader clock / power / temperature while the timed launches were executing
        rec = {
            "metric": "MERL materials fitted/sec" if name.startswith("merl_fit") else "fits/sec" if name.startswith("fit_") else "BRDF evals/sec",
            "value": value, "unit": f"{unit}/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32 (f64 transcendentals)", "data": "synthetic",
            "config": {"workload": name, "units_per_gpu_per_step": n,
                       "brdf": {"merl_eval": "MERL 90x90x180x3 nearest-bin (synthetic GGX0.3+diffuse table)",
                                "ggx_eval_pdf": f"GGX isotropic alpha={args.alpha:g}, {args.fresnel} Fresnel, eval+pdf fused",
                                "beckmann_sample": "Beckmann elliptic(0.2,0.5,0.7) VNDF sample, on-chip RNG",
                                "beckmann_sample_contract": "Beckmann elliptic(0.2,0.5,0.7) VNDF sample, on-chip RNG, DJB_OPT_CONTRACT_1E5 (directions within 1e-5 per component, not bit-identical)",
                                "utia_eval": "UTIA 6x48x6x48x3 table, 16-tap interpolation + sRGB decode (synthetic payload)",
                                "utia_eval_contract": "UTIA 6x48x6x48x3 table, 16-tap interpolation + sRGB decode (synthetic payload), DJB_OPT_CONTRACT_1E5",
                                "merl_fit": "tabular(merl, 90) + fit_beckmann + fit_ggx per material, tables resident in HBM",
                                "merl_fit_files": "files on local disk -> pread -> PCIe -> k_merl_convert -> "
                                                  "tabular(merl, 90) + both fits (end to end)",
                                "ggx_unpolarized_eval_pdf": f"GGX isotropic alpha={args.alpha:g}, unpolarized Fresnel ior (1.5, 1.8, 2.4), eval+pdf fused",
                                "ggx_unpolarized_eval_pdf_contract": f"GGX isotropic alpha={args.alpha:g}, unpolarized Fresnel ior (1.5, 1.8, 2.4), eval+pdf fused, DJB_OPT_CONTRACT_1E5",
                                "sgd_eval": "sgd::eval, gold-metallic-paint (published row)",
                                "sgd_eval_contract": "sgd::eval, gold-metallic-paint, DJB_OPT_CONTRACT_1E5",
                                "ggx_eval_pdf_contract": f"GGX isotropic alpha={args.alpha:g}, {args.fresnel} Fresnel, eval+pdf fused, "
                                                         "DJB_OPT_CONTRACT_1E5 (values within 1e-5 relative of the reference, not bit-identical)",
                                "merl_eval_uniform_bins": "MERL nearest-bin, look-ups uniform over all 90x90x180 bins",
                                "merl_eval_coherent": "MERL nearest-bin, renderer-like coherent batch (bumpy plane, one light)"}.get(name, kernel),
                       "layout": "SoA float32 in HBM", "parallelism": f"independent x{world} (no collective)"
                       + (" -- SELF-TEST batch size (--selftest-n): not the BASELINE configuration" if args.selftest_n else "")
                       + (" -- SELF-TEST: all ranks share GPU 0 (DJB_BENCH_SHARE_GPU), not a scaling measurement" if share_gpu else "")},
            "roofline": roofline,
            "ranks": ranks,
        }
        if world > 1 or name.startswith("merl_fit") or want_secondary:
            rec["scaling_model_ms"] = scaling_model_ms(world)
        if name == "merl_fit_files":
            rec["pipeline"] = keep[1].get("timing")   # last step: total / load (read+upload+convert) / fit seconds
        if world == 1 and not args.no_cpu_baseline:
            rec["cpu_baseline"] = cpu_baseline(name, synth)
        if want_secondary and world > 1:
            rec["secondary"] = {"merl_fit_files_100": fitfiles, "merl_fit_100": fit100}
        if want_secondary and world == 1:
            if not args.no_cpu_baseline:
                fitfiles["cpu_baseline"] = cpu_baseline("merl_fit_files", synth)
            sec = {"merl_fit_files_100": fitfiles, "merl_fit_100": fit100}
            if fitdir is not None:
                sec["merl_fit_dir"] = fitdir
            opc = one_pair_calls(not args.no_cpu_baseline)
            if opc.get("gpu_object_host_twin"):
                sec["one_pair_calls"] = opc
            for other in ("ggx_eval_pdf", "ggx_eval_pdf_contract", "ggx_unpolarized_eval_pdf", "ggx_unpolarized_eval_pdf_contract",
                          "sgd_eval", "sgd_eval_contract", "beckmann_sample", "beckmann_sample_contract", "utia_eval", "utia_eval_contract",
                          "merl_eval_uniform_bins", "merl_eval_coherent"):
                on, ob, ou, _ = WORKLOADS[other]
                if other.startswith("merl_eval_"):
                    on //= 4          # 2.5e8 pairs (9 GB of streams, far beyond every cache): same rate as 1e9, a quarter of the set-up time
                st, kp = make_step(other, on, djb, synth, ctx, torch)
                ms, spread, state = timed_leg(st, torch, local)
                finish(st)
                sec[other] = {"value": on / (ms * 1e-3), "unit": f"{ou}/s", "ms_per_step": ms, "ms_min_median_max": spread, "gpu": state,
                              "units_per_step": on,
                              "hbm_GBps": (on * ob / (ms * 1e-3) / 1e9) if ob else None,
                              "roofline_frac": (on * ob / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if ob else None}
                sec[other].update(static_profile(other, on, ms))
                if other == "beckmann_sample_contract":
                    acc = djb.selftest_contract_sample(kp[2], kp[3], n=1 << 28, seed=3, family=0, ctx=ctx)
                    sec[other].update({"max_abs_err_direction": acc["max_abs_dir"], "components_outside_1e-5": acc["outside_1e5"],
                                       "exact_path_share": acc["exact_path"] / acc["samples"], "error_bound_used": acc["bound_used"],
                                       "contract": "every component of the sampled unit vector within 1e-5 of the reference's; samples whose decisions "
                                                   "or conditioning are in doubt take the bit-exact path in the same launch; DJB_OPT_CONTRACT_1E5, off by default"})
                elif other == "utia_eval_contract":
                    sec[other].update(utia_contract_accuracy(st, kp, djb, ctx, torch))
                elif other.endswith("_contract"):
                    # measured accuracy of the fast path against the bit-exact per-pair code, same set-up, 2^28 generated pairs
                    acc = djb.selftest_contract(kp[2], kp[3], n=1 << 28, seed=3, family=0, ctx=ctx)
                    sec[other].update({"max_rel_err_eval": acc["max_rel_eval"], "max_rel_err_pdf": acc["max_rel_pdf"],
                                       "values_outside_1e-5": acc["outside_1e5"], "zero_pattern_mismatches": acc["zero_mismatch"],
                                       "exact_tier_share": acc["tier2"] / acc["pairs"], "contract": "1e-5 relative (north_star), "
                                       "bit-exact tier for ill-conditioned pairs; DJB_OPT_CONTRACT_1E5, off by default"})
                del st, kp
                torch.cuda.empty_cache()
            # the operators the five plugins issue beyond the BASELINE configs: driver-observed throughput per leg, priced
            # against 8 TB/s with the algorithmic bytes WORKLOADS states; counters per leg: profiles/pmc_<leg>.json, valu_<leg>.json
            ops = {}
            for leg in PLUGIN_LEGS:
                on, ob, ou, kern = WORKLOADS[leg]
                st, kp = make_step(leg, on, djb, synth, ctx, torch)
                fit_leg = leg.startswith("fit_")
                ms, spread, state = timed_leg(st, torch, local, reps=5 if fit_leg else 10, warm=2 if fit_leg else 10)
                ops[leg] = {"value": on / (ms * 1e-3), "unit": f"{ou}/s", "ms_per_step": ms, "ms_min_median_max": spread, "gpu": state,
                            "units_per_step": on, "kernel": kern, "algorithmic_bytes_per_unit": ob,
                            "hbm_GBps": (on * ob / (ms * 1e-3) / 1e9) if ob else None,
                            "roofline_frac": (on * ob / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if ob else None}
                ops[leg].update(static_profile(leg, on, ms))
                del st, kp
                torch.cuda.empty_cache()
            sec["plugin_ops"] = ops
            sec["merl_eval_order_lever"] = merl_order_lever(djb, synth, ctx, torch, local)
            rec["secondary"] = sec
        print(json.dumps(rec), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
