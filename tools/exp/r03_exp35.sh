#!/bin/bash
# round 3, GPU call 35: A/B of the two-instruction near_f32_midpoint against the previous commit's library, same box, alternating
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; O=gpurun_out/r03; mkdir -p $O
one() { timeout 600 python bench.py --workload $1 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2 ms %.3f' % r['ms_per_step'])"; }
for i in 1 2 3; do
  for w in beckmann_sample ggx_eval_pdf; do
    DJB_LIB_PATH=$R/gpurun_variants/libdjb_prev.so one $w prev
    one $w new
  done
done
