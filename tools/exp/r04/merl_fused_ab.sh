#!/bin/bash
# round 4: A/B on one box: two = tier 2 as a second kernel (the committed state before the fusion, gpurun_variants/libdjb_two.so), fused = tier 2 drained in-kernel
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
O=gpurun_out/merl_fused_ab.txt; : > $O
for rep in 1 2 3; do for v in two fused; do
  lib=gpurun_variants/libdjb_$v.so; [ $v = fused ] && lib=dj_brdf_amd/lib/libdjb_hip.so
  for w in merl_eval merl_eval_uniform_bins; do
    A=""; case $w in merl_eval_*) A="--n 250000000";; esac
    DJB_LIB_PATH=$lib timeout 300 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-secondary $A 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('%-6s %-24s %8.3f ms/step  frac %.3f' % ('$v', '$w', d['ms_per_step'], d['roofline']['frac']))" >> $O
  done; done; done
cat $O
