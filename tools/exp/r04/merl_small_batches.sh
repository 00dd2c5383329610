#!/bin/bash
# round 4: the MERL look-up on small batches: four pairs per lane (k_merl_fast_v4) vs one (k_merl_fast), replayed from a graph so that
# the kernel's own latency shows -> profiles/r04/merl_small_batches.txt   (DJB_MERL_V4_MIN: batches below it take the one-pair kernel)
# library: the shipped one built with EXTRA=-DDJB_EXPERIMENT (DJB_MERL_V4_MIN is read only then; shipped threshold 2^18)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
O=gpurun_out/merl_small_batches.txt; : > $O
for m in 0 1000000000; do
  echo "DJB_MERL_V4_MIN=$m" >> $O
  DJB_MERL_V4_MIN=$m timeout 300 python - >> $O 2>/dev/null <<'PY'
import ctypes as C, time, torch
from dj_brdf_amd import djb, synth
ctx = djb.default_context(0); lib = djb._lib.load(); K = 128
obj = djb.merl.from_table(synth.merl_table(0.3), ctx=ctx)
for lg in range(10, 23):
    n = 1 << lg
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        i = djb.gen_directions(n, synth.SEED_I, ctx=ctx); o = djb.gen_directions(n, synth.SEED_O, ctx=ctx)
        vi, vo = djb._Vec(i), djb._Vec(o); out = torch.zeros((3, n), dtype=torch.float32, device=i.device); vout = djb._Vec(out)
        call = lambda: lib.djb_eval_batch(ctx._h, obj._h, C.c_int64(n), C.byref(vi.view), C.byref(vo.view), None, C.byref(vout.view), C.c_int(0))
        for _ in range(3): call()
        side.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        for _ in range(K): call()
    g.replay(); torch.cuda.synchronize()
    ts = []
    for _ in range(15):
        t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print("n = 2^%-2d  %8.2f us per call" % (lg, sorted(ts)[7] / K * 1e6))
PY
done
cat $O
