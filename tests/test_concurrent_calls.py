"""Several host threads on ONE context (INTEGRATION.md "Ownership and threading": Mitsuba's render threads share a BSDF): batch
operators on host arrays, fits from files (djb_fit_merl_files: parked reader threads, a staging buffer and slot plans that stay
with the context) and fits of objects, issued concurrently, must return what the same calls return one after the other."""
import os
import threading

import numpy as np
import pytest

from dj_brdf_amd import djb, merl_params, synth


def _bits(a):
    return np.ascontiguousarray(np.asarray(a, np.float32)).view(np.uint32)


def _exercise(ctx, tmp_path):
    files = []
    for k, a in enumerate((0.3, 0.1, 0.05, 0.6, 0.2, 0.4)):
        p = os.path.join(str(tmp_path), f"m{k}.binary"); synth.write_merl_binary(p, synth.merl_table(alpha=a)); files.append(p)
    P = djb.microfacet.params
    g = djb.ggx(djb.fresnel.schlick((0.9, 0.6, 0.3)), True, ctx=ctx)
    m = djb.merl.from_table(synth.merl_table_hashed(), ctx=ctx)
    t = djb.tabular(g, 40, True, ctx=ctx)
    n = 50_000

    def job(k):
        i, o = synth.directions_aos(n, 100 + k), synth.directions_aos(n, 200 + k)
        p = P.elliptic(0.1 + 0.05 * k, 0.5, 0.3)
        sel = files[k % 3:] + files[:k % 3]
        ab, ag, _ = merl_params.fit_files_on(ctx, sel)
        tt = djb.tabular(m, 20 + k, True, ctx=ctx)
        r = [g.eval(i, o, p), g.pdf(i, o, p), m.eval(i, o), t.eval(i, o, p), t.sample(synth.uniforms(n, 300 + k), synth.uniforms(n, 400 + k), o, p),
             np.asarray(ab), np.asarray(ag), tt.get_p22v(), np.asarray(djb.tabular.fit_ggx_parameters(tt).get_ellipse()[:1])]
        tt.close()
        return r

    want = [job(k) for k in range(8)]
    got = [None] * 8
    err = []

    def work(k):
        try:
            for _ in range(3):
                got[k] = job(k)
        except Exception as e:          # noqa: BLE001 -- reported below, in the main thread
            err.append((k, repr(e)))
    th = [threading.Thread(target=work, args=(k,)) for k in range(8)]
    [x.start() for x in th]; [x.join() for x in th]
    assert not err, err
    for k in range(8):
        for a, b in zip(got[k], want[k]):
            assert np.array_equal(_bits(a), _bits(b)), f"thread {k}: a concurrent call returned other bits than the sequential one"


def test_eight_threads_on_one_cpu_context(tmp_path):
    _exercise(djb.Context("cpu"), tmp_path)


@pytest.mark.gpu
def test_eight_threads_on_one_gpu_context(gpu_ctx, tmp_path):
    _exercise(gpu_ctx, tmp_path)
