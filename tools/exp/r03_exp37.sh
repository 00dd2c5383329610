#!/bin/bash
# round 3, GPU call 37: what IEEE fp32 division costs the exact GGX eval kernel: timing-only build with the 2.5-ulp division
# (-fno-hip-fp32-correctly-rounded-divide-sqrt; results differ, not shipped) against the shipped library
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"
one() { timeout 600 python bench.py --workload $1 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2 ms %.3f' % r['ms_per_step'])"; }
for i in 1 2; do
  for w in ggx_eval_pdf beckmann_sample; do
    one $w shipped
    DJB_LIB_PATH=$R/gpurun_variants/libdjb_fd.so one $w fastdiv
  done
done
