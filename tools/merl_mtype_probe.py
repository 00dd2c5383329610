#!/usr/bin/env python3
"""Does the memory type of the MERL kernel's 36 B/pair streams matter?  Inputs / outputs allocated with
hipExtMallocWithFlags (default, fine-grained, uncached) instead of hipMalloc, same kernel, bench distribution.
    PYTHONPATH=. python tools/merl_mtype_probe.py > profiles/r04/merl_mtype_probe.txt"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dj_brdf_amd import djb, synth, _lib
ctx = djb.default_context(0); lib = _lib.load()
hip = C.CDLL("libamdhip64.so")
n = 500_000_000
FL = {"default": 0, "finegrained": 1, "uncached": 3}
def alloc(kind):
    p = C.c_void_p()
    if kind == "hipMalloc":
        rc = hip.hipMalloc(C.byref(p), C.c_size_t(12 * n))
    else:
        rc = hip.hipExtMallocWithFlags(C.byref(p), C.c_size_t(12 * n), C.c_uint(FL[kind]))
    assert rc == 0, (kind, rc)
    return p.value
def view(base):
    v = _lib.Vec3View(); v.x, v.y, v.z, v.stride = base, base + 4 * n, base + 8 * n, 1
    return v
m = djb.merl.from_table(synth.merl_table(0.3), ctx=ctx)
CONFIGS = (("hipMalloc", "hipMalloc"), ("uncached", "hipMalloc"), ("hipMalloc", "uncached"), ("uncached", "uncached"),
           ("finegrained", "finegrained"), ("hipMalloc", "hipMalloc"))
if os.environ.get("DJB_MTYPE_ONLY"):      # one configuration only, e.g. "uncached,uncached": a run under rocprofv3 --pmc (round 5)
    CONFIGS = (tuple(os.environ["DJB_MTYPE_ONLY"].split(",")),)
for kin, kout in CONFIGS:
    pi, po, pr = alloc(kin), alloc(kin), alloc(kout)
    vi, vo, vr = view(pi), view(po), view(pr)
    _lib.check(lib.djb_gen_directions(ctx._h, C.c_int64(n), C.c_uint32(synth.SEED_I), C.c_uint64(0), C.byref(vi)))
    _lib.check(lib.djb_gen_directions(ctx._h, C.c_int64(n), C.c_uint32(synth.SEED_O), C.c_uint64(0), C.byref(vo)))
    def step():
        _lib.check(lib.djb_eval_batch(ctx._h, m._h, C.c_int64(n), C.byref(vi), C.byref(vo), None, C.byref(vr), C.c_int(0)))
    for _ in range(4): step()
    torch.cuda.synchronize(); ctx.timer_start()
    for _ in range(8): step()
    ms = ctx.timer_stop_ms() / 8
    chk = torch.empty(0)
    print(f"inputs {kin:11s} outputs {kout:11s}: {ms:7.3f} ms per 5e8 pairs = {36 * n / ms / 1e6 / 8000:.3f} of 8 TB/s", flush=True)
    for p in (pi, po, pr): hip.hipFree(C.c_void_p(p))
