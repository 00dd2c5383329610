#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
FIT_RATES_SPARSE_ONLY=1 DJB_LOADER_TRACE=0 PYTHONPATH=. timeout 600 python tools/fit_files_rates.py > gpurun_out/fit_files_rates.txt 2>&1; cat gpurun_out/fit_files_rates.txt
DJB_LOADER_TRACE=1 PYTHONPATH=. timeout 300 python - 2>&1 <<'PY' | tail -12
import bench, time
from dj_brdf_amd import djb, merl_params, synth
ctx = djb.Context(0)
paths = bench.synth_merl_files(100, synth)
for _ in range(4):
    time.sleep(0.1); merl_params.fit_files_on(ctx, paths)
PY
