#!/usr/bin/env python3
"""End-to-end 100-file MERL fit (files -> alphas) for several reader-thread counts (run on the GPU box).
PYTHONPATH=. python tools/fit_files_rates.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from dj_brdf_amd import djb, merl_params, synth
ctx = djb.Context(0)
paths = bench.synth_merl_files(100, synth)
merl_params.fit_files_on(ctx, paths[:4])
legs = ((False, (8, 16, 25, 32, 50, 64, 100)),) if os.environ.get("FIT_RATES_SPARSE_ONLY") else ((False, (1, 2, 4, 8, 16, 32)), (True, (2, 4, 8)))
print("DJB_GATHER_MODE =", os.environ.get("DJB_GATHER_MODE", "(default)"))
for dense, sweep in legs:
  djb.set_fit_files_dense(ctx, dense)
  print("dense upload of whole tables" if dense else "sparse: only the entries the fit reads")
  for threads in sweep:
    best = None
    for rep in range(3):
        time.sleep(0.08)      # let the previous call's reaper thread finish releasing its mappings
        t0 = time.perf_counter()
        ab, ag, tim = merl_params.fit_files_on(ctx, paths, reader_threads=threads)
        wall = time.perf_counter() - t0
        if best is None or wall < best[0]: best = (wall, tim)
    print(f"  threads {threads:2d}: wall {best[0]*1e3:7.1f} ms  (pipeline total {best[1]['total_s']*1e3:.1f}, load {best[1]['load_s']*1e3:.1f}, fit {best[1]['fit_s']*1e3:.2f} ms)  "
          f"{best[1]['bytes']/best[1]['load_s']/1e9:.1f} GB/s", flush=True)
