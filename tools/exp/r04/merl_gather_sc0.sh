#!/bin/bash
# round 4: every table gather of the MERL look-up with sc0 (skip the CU's 32 KB vector cache, which a 17.5 MB table never hits) --
# variant library gpurun_variants/libdjb_gpol.so (the exec-masked asm gathers of commit 40cea38 + policy 4 = sc0); hot_rows 0 = every row
# takes the policy, 8100 = none (the asm path with plain loads: the control) -> profiles/r04/merl_gather_sc0.txt
# variant: git diff bc0b574 40cea38 -- dj_brdf_amd/csrc/djb_kernels_merl.hip applied (+ djb_merl_row_rank.inc of 40cea38, + `else if (policy == 4) DJB_MERL_GATHER_ASM("sc0")`), make BUILD=build_gpol OUT=../../gpurun_variants/libdjb_gpol.so
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
O=gpurun_out/merl_gather_sc0.txt; : > $O
run() { # label lib policy rows workload
  A=""; case $5 in merl_eval_*) A="--n 250000000";; esac
  DJB_LIB_PATH=$2 DJB_MERL_COLD_POLICY=$3 DJB_MERL_HOT_ROWS=$4 timeout 300 python bench.py --workload $5 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary $A 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('%-28s %-24s %8.3f ms/step  frac %.3f' % ('$1', '$5', d['ms_per_step'], d['roofline']['frac']))" >> $O
}
V=gpurun_variants/libdjb_gpol.so; S=dj_brdf_amd/lib/libdjb_hip.so
for rep in 1 2 3; do
  for w in merl_eval merl_eval_uniform_bins merl_eval_coherent; do
    run "shipped" $S 0 0 $w
    run "asm gathers, plain" $V 1 8100 $w
    run "asm gathers, all sc0" $V 4 0 $w
    run "asm gathers, all sc1" $V 2 0 $w
  done
done
cat $O
