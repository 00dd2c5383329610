"""Device-memory batch calls are asynchronous launches on the context's stream and nothing else (no allocation, no synchronisation,
no host read-back for batches below 2^20 units once a kind has run once on the context), so a caller can capture a launch-bound inner
loop -- many small batches, as a renderer's wavefront queue produces them -- into a hipGraph and replay it.  Checked here through
torch's graph capture: the replayed results equal the eager ones bit for bit, for every two-tier / table / sampler family."""
import ctypes as C

import pytest

from dj_brdf_amd import djb, synth

pytestmark = pytest.mark.gpu


def _calls(ctx, torch, n):
    """[(name, launch(), out tensors)] over preallocated device buffers"""
    lib = djb._lib.load()
    dev = f"cuda:{ctx.device}"
    i = djb.gen_directions(n, synth.SEED_I, ctx=ctx); o = djb.gen_directions(n, synth.SEED_O, ctx=ctx)
    vi, vo = djb._Vec(i), djb._Vec(o)
    keep = [i, o, vi, vo]
    calls = []

    par = djb._params_ptr

    def add_eval_pdf(name, obj, p):
        fr = torch.zeros((3, n), dtype=torch.float32, device=dev); pdf = torch.zeros(n, dtype=torch.float32, device=dev)
        vfr = djb._Vec(fr); keep.extend([obj, p, vfr])
        calls.append((name, lambda: djb._lib.check(lib.djb_eval_pdf_batch(ctx._h, obj._h, C.c_int64(n), C.byref(vi.view), C.byref(vo.view), par(p), C.c_int(0),
                                                                         C.byref(vfr.view), C.c_void_p(pdf.data_ptr()), C.c_int(0))), (fr, pdf)))

    def add_eval(name, obj):
        fr = torch.zeros((3, n), dtype=torch.float32, device=dev)
        vfr = djb._Vec(fr); keep.extend([obj, vfr])
        calls.append((name, lambda: djb._lib.check(lib.djb_eval_batch(ctx._h, obj._h, C.c_int64(n), C.byref(vi.view), C.byref(vo.view), None,
                                                                     C.byref(vfr.view), C.c_int(0))), (fr,)))

    add_eval_pdf("ggx eval+pdf", djb.ggx(djb.fresnel.schlick((1.0, 0.71, 0.29)), True, ctx=ctx), djb.microfacet.params.isotropic(0.3))
    add_eval_pdf("beckmann eval+pdf", djb.beckmann(ctx=ctx), djb.microfacet.params.elliptic(0.2, 0.5, 0.7))
    add_eval("merl eval", djb.merl.from_table(synth.merl_table_hashed(), ctx=ctx))
    add_eval("utia eval", djb.utia.from_table(synth.utia_table_smooth(), ctx=ctx))
    add_eval("sgd eval", djb.sgd("gold-metallic-paint", ctx=ctx))
    # Beckmann VNDF sample with on-chip uniforms
    bk = djb.beckmann(ctx=ctx); pe = djb.microfacet.params.elliptic(0.2, 0.5, 0.7)
    si = torch.zeros((3, n), dtype=torch.float32, device=dev); vsi = djb._Vec(si); keep.extend([bk, pe, vsi])
    calls.append(("beckmann sample (rng)", lambda: djb._lib.check(lib.djb_sample_rng_batch(ctx._h, bk._h, C.c_int64(n), C.c_uint32(synth.SEED_U1), C.c_uint32(synth.SEED_U2),
                                                                                          C.c_uint64(0), C.byref(vo.view), par(pe), C.byref(vsi.view))), (si,)))
    return calls, keep


def test_small_batches_replay_from_a_captured_graph(gpu_ctx):
    import torch
    n = 1 << 16
    side = torch.cuda.Stream(device=gpu_ctx.device)
    with torch.cuda.stream(side):           # the context follows torch's current stream
        calls, keep = _calls(gpu_ctx, torch, n)
        for _, launch, _ in calls:          # eager: the reference results (also the one call per kind that may allocate)
            launch()
        side.synchronize()
        want = [[t.clone() for t in outs] for _, _, outs in calls]
        for _, _, outs in calls:
            for t in outs:
                t.zero_()
        side.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        for _, launch, _ in calls:
            launch()
    for _, _, outs in calls:                # capture executes nothing
        for t in outs:
            assert not t.any(), "a call ran during capture instead of being recorded"
    g.replay()
    torch.cuda.synchronize()
    for (name, _, outs), w in zip(calls, want):
        for t, e in zip(outs, w):
            assert torch.equal(t.view(torch.int32), e.view(torch.int32)), f"{name}: graph replay differs from the eager call"
    # and again: a graph is replayable
    for _, _, outs in calls:
        for t in outs:
            t.zero_()
    g.replay()
    torch.cuda.synchronize()
    for (name, _, outs), w in zip(calls, want):
        for t, e in zip(outs, w):
            assert torch.equal(t.view(torch.int32), e.view(torch.int32)), f"{name}: second replay differs"
