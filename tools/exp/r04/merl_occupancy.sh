#!/bin/bash
# round 4: tier 1 of the MERL lookup at a forced occupancy (-DDJB_EXP_MERL_WAVES=n variants: amdgpu_waves_per_eu(n, n) on k_merl_fast_v4) --
# would the kernel tolerate the exact path's 143 VGPRs (3 waves per SIMD) if tier 2 were drained inside it?
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
O=gpurun_out/merl_occupancy.txt; : > $O
for rep in 1 2; do for v in base w4 w3 w2; do
  lib=gpurun_variants/libdjb_$v.so; [ $v = base ] && lib=dj_brdf_amd/lib/libdjb_hip.so
  for w in merl_eval merl_eval_coherent; do
    A=""; case $w in merl_eval_*) A="--n 250000000";; esac
    DJB_LIB_PATH=$lib timeout 300 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-secondary $A 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('%-6s %-20s %8.3f ms/step' % ('$v', '$w', d['ms_per_step']))" >> $O
  done; done; done
cat $O
