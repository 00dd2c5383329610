#!/bin/bash
# round 5: what the deferred samples cost -- the sampler with nobody deferred (-DDJB_EXP_NO_DEFER: timing only, wrong values for the flagged
# samples) against the shipped library, exact and contract mode
# builds: make -C dj_brdf_amd/csrc BUILD=build_nodefer OUT=../../gpurun_variants/libdjb_nodefer.so EXTRA=-DDJB_EXP_NO_DEFER (flags dead too)
#         make -C dj_brdf_amd/csrc BUILD=build_nodefer2 OUT=../../gpurun_variants/libdjb_nodefer2.so EXTRA=-DDJB_EXP_NO_DEFER=2 (flags computed, drains never run)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
O=gpurun_out/beckmann_defer_cost.txt; : > $O
for rep in 1 2; do for v in shipped nodefer nodefer2; do
  lib=gpurun_variants/libdjb_$v.so; [ $v = shipped ] && lib=dj_brdf_amd/lib/libdjb_hip.so
  for w in beckmann_sample beckmann_sample_contract; do
    DJB_LIB_PATH=$lib timeout 300 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('%-8s %-26s %8.3f ms/step  frac %.3f' % ('$v', '$w', d['ms_per_step'], d['roofline']['frac']))" >> $O
  done; done; done
cat $O
