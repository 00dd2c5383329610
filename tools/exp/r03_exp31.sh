#!/bin/bash
# round 3, GPU call 31: k_fit sigma producers, rows held in registers: same-box A/B of the compute-only fit (100 / 13 / 1 materials) + phase stamps
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; O=gpurun_out/r03; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "fit or tabular or pipeline" 2>&1 | tail -2
DJB_LIB_PATH=gpurun_variants/libdjb_ts.so PYTHONPATH=. timeout 300 python - > $O/fit_phases3.txt 2>&1 <<'PY'
from dj_brdf_amd import djb, synth
ctx = djb.Context(0)
for n in (100, 13, 1):
    mats = [djb.merl.from_table(synth.merl_table(*synth.material_recipe(k)), ctx=ctx) for k in range(n)]
    for rep in range(3):
        djb.fit_brdf_batch(mats, 90, True, ctx=ctx)
PY
grep djb_exp $O/fit_phases3.txt | awk 'NR%3==0'
for rep in 1 2; do for v in prev new; do
  lib=$([ $v = new ] && echo "" || echo gpurun_variants/libdjb_prev.so)
  for n in 100 13 1; do
    DJB_LIB_PATH=$lib timeout 300 python bench.py --workload merl_fit --n $n --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$v n=$n: %.4f ms' % r['ms_per_step'])"
  done
done; done
