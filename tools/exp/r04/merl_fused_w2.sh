#!/bin/bash
# round 4: the fused MERL kernel at 2 waves per SIMD (variant: make BUILD=build_w2 OUT=../../gpurun_variants/libdjb_w2.so EXTRA=-DDJB_EXP_MERL_WAVES=2) -> appended to profiles/r04/merl_occupancy.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for rep in 1 2 3; do for v in base w2; do lib=dj_brdf_amd/lib/libdjb_hip.so; [ $v = w2 ] && lib=gpurun_variants/libdjb_w2.so
  for w in merl_eval merl_eval_uniform_bins; do A=""; case $w in merl_eval_*) A="--n 250000000";; esac
  DJB_LIB_PATH=$lib timeout 300 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-secondary $A 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$v %-24s %8.3f ms' % ('$w', d['ms_per_step']))"; done; done; done
