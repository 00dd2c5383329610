#!/bin/bash
# round 3, GPU call 32: final k_fit of the round: phase stamps, rocprofv3 passes of the merl_fit leg, full GPU suite
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; O=gpurun_out/r03; mkdir -p $O
DJB_LIB_PATH=gpurun_variants/libdjb_ts.so PYTHONPATH=. timeout 300 python - > $O/fit_phases_final.txt 2>&1 <<'PY'
from dj_brdf_amd import djb, synth
ctx = djb.Context(0)
for n in (100, 13, 1):
    mats = [djb.merl.from_table(synth.merl_table(*synth.material_recipe(k)), ctx=ctx) for k in range(n)]
    for rep in range(3):
        djb.fit_brdf_batch(mats, 90, True, ctx=ctx)
PY
grep djb_exp $O/fit_phases_final.txt | awk 'NR%3==0'
WORKLOADS=merl_fit timeout 600 bash tools/profile_bench.sh > gpurun_out/profile_bench_fit.log 2>&1; tail -1 gpurun_out/prof/merl_fit/bench_plain.json | cut -c1-260
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
