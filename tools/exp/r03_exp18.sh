#!/bin/bash
# round 3, GPU call 18: Beckmann sample two-path kernel: rolled vs unrolled trip loop, evalp_is rates, previous commit's one-kernel form as baseline
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; O=gpurun_out/r03; mkdir -p $O
B="python bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-secondary"
for v in ship un; do
  lib=$([ $v = ship ] && echo "" || echo gpurun_variants/libdjb_$v.so)
  for rep in 1 2; do
    DJB_LIB_PATH=$lib timeout 300 $B --workload beckmann_sample > $O/bk4_${v}_$rep.json 2>/dev/null
    python -c "import json;print('$v 1e9', '%.3f' % json.loads(open('$O/bk4_${v}_$rep.json').read().strip().splitlines()[-1])['ms_per_step'])"
  done
  echo "== sample_rates $v"; DJB_LIB_PATH=$lib PYTHONPATH=. timeout 300 python tools/sample_rates.py 2>/dev/null
done
echo "== sample_rates one-kernel form (previous commit)"; DJB_LIB_PATH=gpurun_variants/libdjb_prev.so PYTHONPATH=. timeout 300 python tools/sample_rates.py 2>/dev/null
