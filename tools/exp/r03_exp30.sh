#!/bin/bash
# round 3, GPU call 30: k_fit sigma producers with their rows' sin / cos held in registers: fit tests, phase stamps, compute-only rate
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; O=gpurun_out/r03; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q -k "fit or tabular or pipeline or aniso or models" 2>&1 | tail -3
DJB_LIB_PATH=gpurun_variants/libdjb_ts.so PYTHONPATH=. timeout 300 python - > $O/fit_phases2.txt 2>&1 <<'PY'
from dj_brdf_amd import djb, synth
ctx = djb.Context(0)
for n in (100, 13, 1):
    mats = [djb.merl.from_table(synth.merl_table(*synth.material_recipe(k)), ctx=ctx) for k in range(n)]
    for rep in range(3):
        djb.fit_brdf_batch(mats, 90, True, ctx=ctx)
PY
grep djb_exp $O/fit_phases2.txt | awk 'NR%3==0'
for r in 1 2 3; do timeout 300 python bench.py --workload merl_fit --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('merl_fit 100: %.4f ms' % r['ms_per_step'])"; done
