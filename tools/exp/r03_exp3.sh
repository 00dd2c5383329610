#!/bin/bash
# round 3, GPU call 3: VALU cost probe, MERL grid sweep, ambiguous share of the uniform-bins leg, the whole -m gpu suite
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; O=gpurun_out/r03; mkdir -p $O
tools/bin/valu_cost_probe > $O/valu_cost.txt 2>&1
B="python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-secondary"
for cap in 4096 16384 65536 262144 100000000; do
  DJB_LIB_PATH=gpurun_variants/libdjb_exp.so DJB_MERL_GRID_CAP_ENV=$cap timeout 300 $B --workload merl_eval > $O/merlgrid_$cap.json 2>$O/merlgrid_$cap.err
done
python - > $O/uniform_guard_stats.txt 2>&1 <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import torch, bench
from dj_brdf_amd import djb, synth
ctx = djb.default_context(0)
for name in ("merl_eval_uniform_bins", "merl_eval_coherent"):
    i, o = bench.merl_pairs(name, 1 << 26, djb, torch, ctx)
    print(name, djb.merl_guard_stats(i, o, ctx=ctx))
i = djb.gen_directions(1 << 26, synth.SEED_I, ctx=ctx); o = djb.gen_directions(1 << 26, synth.SEED_O, ctx=ctx)
print("bench distribution", djb.merl_guard_stats(i, o, ctx=ctx))
PY
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gpu_suite.txt 2>&1; echo "gpu suite rc=$?" >> $O/gpu_suite.txt
cat $O/valu_cost.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03/merlgrid_*.json")):
    try:
        r=json.loads(open(f).read().strip().splitlines()[-1])
        print("%-44s %8.3f ms  %8.2f G/s  frac %.3f" % (f.split('/')[-1], r["ms_per_step"], r["value"]/1e9, r["roofline"]["frac"] or 0))
    except Exception as e: print(f, "ERR", e)
PY
cat $O/uniform_guard_stats.txt; tail -5 $O/gpu_suite.txt
