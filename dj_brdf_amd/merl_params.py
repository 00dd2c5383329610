"""Batch counterpart of the reference's example driver (examples/merl_params.cpp:30-73):

    python -m dj_brdf_amd.merl_params [-o params.txt] [--gpus N] a.binary b.binary ...

fits Beckmann / GGX roughness to every MERL file with the HIP power-iteration kernel (``--cpu``, or a machine
without a HIP device: the library's host path) and writes
the same ``params.txt`` ("# MERL Beckmann GGX" then ``name %.3f %.3f`` per file, input order).
Per GPU the work is the native pipeline of ``djb_fit_merl_files`` (reader threads -> pinned ring ->
async upload + conversion kernel -> one fit launch).  Materials are independent: with several GPUs
visible they are dealt round-robin, one host thread + one HIP stream per GPU, no collective
(SURVEY.md 8e).
"""
from __future__ import annotations

import argparse
import ctypes as C
import os
import sys
import threading
import time

import numpy as np

from . import _lib, djb, shard


def material_name(path: str) -> str:
    """sscanf(strrchr(input, '/') + 1, "%[^.]s", name): basename up to the first '.'
    (examples/merl_params.cpp:64); at most 63 characters fit the reference's char name[64]."""
    return os.path.basename(path).split(".")[0][:63]


def read_merl_payload(path: str) -> np.ndarray:
    """The 3*n doubles of a MERL file (header checked like dj_brdf.h:963-983)."""
    try:
        with open(path, "rb") as f:
            dims = np.fromfile(f, dtype=np.int32, count=3)
            n = int(np.prod(dims.astype(np.int64))) if dims.size == 3 else 0
            if n <= 0:
                raise djb.exc(3, "djb_error: Failed to read MERL header\n")
            data = np.fromfile(f, dtype=np.float64, count=3 * n)
    except OSError:
        raise djb.exc(2, f"djb_error: Failed to open {path}\n")
    if data.size != 3 * n:
        raise djb.exc(4, f"djb_error: Reading {path} failed\n")
    return data


def fit_files_on(ctx: djb.Context, paths, res=90, shadow=True, reader_threads=0):
    """(alpha_beckmann[n], alpha_ggx[n], timing dict) for `paths` on one GPU (native pipeline)."""
    lib = _lib.load()
    n = len(paths)
    arr = (C.c_char_p * max(n, 1))(*[p.encode() for p in paths])
    ab, ag = np.zeros(n, np.float32), np.zeros(n, np.float32)
    timing = (C.c_double * 4)()
    _lib.check(lib.djb_fit_merl_files(ctx._h, C.c_int(n), arr, C.c_int(res), C.c_int(int(shadow)),
                                      C.c_int(reader_threads), C.c_void_p(ab.ctypes.data),
                                      C.c_void_p(ag.ctypes.data), timing))
    return ab, ag, {"total_s": timing[0], "load_s": timing[1], "fit_s": timing[2], "bytes": timing[3]}


def fit_files_multi(ctxs, paths, res=90, shadow=True, reader_threads=0):
    """(alpha_beckmann[n], alpha_ggx[n], [timing dict per context]) for `paths` over several contexts: djb_fit_merl_files_multi --
    file k on context k mod len(ctxs), one host thread per context INSIDE the library, rows in input order, no exchange."""
    lib = _lib.load()
    n = len(paths)
    arr = (C.c_char_p * max(n, 1))(*[p.encode() for p in paths])
    handles = (C.c_void_p * len(ctxs))(*[c._h for c in ctxs])
    ab, ag = np.zeros(n, np.float32), np.zeros(n, np.float32)
    timing = (C.c_double * (4 * len(ctxs)))()
    _lib.check(lib.djb_fit_merl_files_multi(handles, C.c_int(len(ctxs)), C.c_int(n), arr, C.c_int(res), C.c_int(int(shadow)),
                                            C.c_int(reader_threads), C.c_void_p(ab.ctypes.data), C.c_void_p(ag.ctypes.data), timing))
    per = [{"total_s": timing[4 * g], "load_s": timing[4 * g + 1], "fit_s": timing[4 * g + 2], "bytes": timing[4 * g + 3]} for g in range(len(ctxs))]
    return ab, ag, per


def fit_files(paths, res=90, shadow=True, gpus=None, return_timing=False, cpu=False):
    """[(alpha_beckmann, alpha_ggx)] for every path, in input order, over `gpus` GPUs -- or, with cpu=True / on a
    machine without a HIP device, on the library's host path (one CPU context, the files spread over its threads):
    the reference's example driver runs without a GPU too (BASELINE configs[0]).  The multi-GPU job is ONE library call
    (djb_fit_merl_files_multi: material m -> GPU m mod G, a host thread per context inside the library)."""
    n_dev = djb.device_count()
    t0 = time.perf_counter()
    if cpu or n_dev == 0:
        ctxs = [djb.Context("cpu")]
    else:
        gpus = min(gpus or n_dev, n_dev, max(len(paths), 1))
        ctxs = [djb.Context(r) for r in range(gpus)]
    ab, ag, timings = fit_files_multi(ctxs, list(paths), res, shadow)
    out = [(float(a), float(g)) for a, g in zip(ab, ag)]
    if return_timing:
        return out, {"wall_s": time.perf_counter() - t0, "per_gpu": timings, "gpus": 0 if (cpu or n_dev == 0) else len(ctxs)}
    return out


def format_params_txt(paths, alphas) -> str:
    lines = ["# MERL Beckmann GGX\n"]
    for p, (ab, ag) in zip(paths, alphas):
        lines.append("%s %.3f %.3f\n" % (material_name(p), ab, ag))
    return "".join(lines)


def main(argv=None):
    ap = argparse.ArgumentParser(prog="merl_params", description="GGX and Beckmann Parameters for MERL BRDFs")
    ap.add_argument("files", nargs="*")
    ap.add_argument("-o", "--output", default="params.txt")
    ap.add_argument("--gpus", type=int, default=None)
    ap.add_argument("--cpu", action="store_true", help="run on the library's host path (default when no HIP device is present)")
    ap.add_argument("--res", type=int, default=90)
    ap.add_argument("--timing", action="store_true", help="print the pipeline timing to stderr")
    args = ap.parse_args(argv)
    if not args.files:
        ap.print_usage()
        return 0
    alphas, timing = fit_files(args.files, res=args.res, gpus=args.gpus, return_timing=True, cpu=args.cpu)
    with open(args.output, "w") as f:
        f.write(format_params_txt(args.files, alphas))
    if args.timing:
        print(timing, file=sys.stderr)
    return 0


if __name__ == "__main__":
    sys.exit(main())
