"""Object lifecycle on a GPU context: every kind of object created, used and destroyed many times must give its HBM back
(djb_brdf_destroy / djb_ctx_destroy), and contexts must be creatable and destroyable in a loop.  HBM is read with
hipMemGetInfo (torch.cuda.mem_get_info) after a warm-up round -- allocator pools, the fit's slot plans and the
context's scratch buffers are allowed to stay -- and again after the measured rounds."""
import gc

import numpy as np
import pytest

from dj_brdf_amd import djb, synth

import os

# hipMemGetInfo counts the whole device: with pytest-xdist the other workers' allocations land in the difference
pytestmark = [pytest.mark.gpu, pytest.mark.skipif("PYTEST_XDIST_WORKER" in os.environ,
                                                  reason="device-wide free-memory readings need the GPU to themselves (run without -n)")]


class _phong(djb.user_brdf):
    def eval(self, i, o, user_param=None):
        c = np.maximum(-o[:, 0] * i[:, 0] - o[:, 1] * i[:, 1] + o[:, 2] * i[:, 2], 0.0)
        return np.repeat((0.1 / np.pi + 0.5 * c ** 20)[:, None], 3, 1).astype(np.float32)


def _one_round(ctx, merl_tab, utia_tab):
    P = djb.microfacet.params
    i, o = synth.directions_aos(4096, 1), synth.directions_aos(4096, 2)
    objs = [djb.ggx(djb.fresnel.schlick((0.9, 0.6, 0.3)), True, ctx=ctx), djb.beckmann(djb.fresnel.unpolarized((1.5, 1.6, 1.7)), ctx=ctx),
            djb.sgd("gold-metallic-paint", ctx=ctx), djb.abc("chrome", ctx=ctx), djb.lambert(ctx=ctx)]
    m = djb.merl.from_table(merl_tab, ctx=ctx); u = djb.utia.from_table(utia_tab, ctx=ctx)
    t = djb.tabular(m, 90, True, ctx=ctx)
    t2 = djb.tabular(objs[2], 33, False, ctx=ctx)
    ta = djb.tabular_anisotropic(u, 10, 14, True, ctx=ctx)
    tu = djb.tabular(_phong(ctx=ctx), 48, True, ctx=ctx)
    bs = djb.beckmann(t.get_fresnel(), ctx=ctx)            # a spline Fresnel term handed on, as the plugins do
    for b in objs + [m, u, t, t2, ta, tu, bs]:
        b.eval(i, o)
    for b in (objs[0], objs[1], t, ta, bs):
        b.pdf(i, o, P.elliptic(0.2, 0.5, 0.7)); b.sample(synth.uniforms(4096, 3), synth.uniforms(4096, 4), o)
    djb.tabular.fit_ggx_parameters(t); djb.tabular_anisotropic.fit_beckmann_parameters(ta)
    for b in objs + [m, u, t, t2, ta, tu, bs]:
        b.close()


def test_objects_give_their_hbm_back(gpu_ctx):
    import torch
    merl_tab, utia_tab = synth.merl_table(), synth.utia_table_smooth()
    for _ in range(3): _one_round(gpu_ctx, merl_tab, utia_tab)
    gc.collect(); torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    rounds = 40
    for _ in range(rounds): _one_round(gpu_ctx, merl_tab, utia_tab)
    gc.collect(); torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    # one round allocates ~30 MB (a MERL table alone is 17.5 MB): a leak of any object would show as hundreds of MB
    assert free0 - free1 < 8 << 20, f"{(free0 - free1) / 2**20:.1f} MiB of HBM not returned after {rounds} rounds"


def test_contexts_can_be_created_and_destroyed_in_a_loop():
    import torch
    merl_tab, utia_tab = synth.merl_table(), synth.utia_table_smooth()
    for _ in range(2):
        c = djb.Context(0); _one_round(c, merl_tab, utia_tab); c.close()
    gc.collect(); torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(12):
        c = djb.Context(0); _one_round(c, merl_tab, utia_tab); c.close()
    gc.collect(); torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < 8 << 20, f"{(free0 - free1) / 2**20:.1f} MiB of HBM not returned after 12 contexts"
