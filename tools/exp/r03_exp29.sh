#!/bin/bash
# round 3, GPU call 29: where k_fit spends its time (phase stamps, DJB_EXP_FIT_TS build) for 100 / 13 / 1 materials; tabular sample rates
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; O=gpurun_out/r03; mkdir -p $O
DJB_LIB_PATH=gpurun_variants/libdjb_ts.so PYTHONPATH=. timeout 300 python - > $O/fit_phases.txt 2>&1 <<'PY'
import sys
from dj_brdf_amd import djb, synth
ctx = djb.Context(0)
for n in (100, 13, 1):
    mats = [djb.merl.from_table(synth.merl_table(*synth.material_recipe(k)), ctx=ctx) for k in range(n)]
    for rep in range(3):
        djb.fit_brdf_batch(mats, 90, True, ctx=ctx)
PY
grep djb_exp $O/fit_phases.txt | awk 'NR%3==0'
DJB_SAMPLE_RATES_TABULAR=1 PYTHONPATH=. timeout 300 python tools/sample_rates.py 2>/dev/null | grep -v amdgpu
