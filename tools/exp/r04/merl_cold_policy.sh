#!/bin/bash
# round 4: table gathers of the rows outside the hottest DJB_MERL_HOT_ROWS (prior: uniform directions) with a streaming cache policy
# (DJB_MERL_COLD_POLICY 0 = all plain, 1 = nt, 2 = sc1, 3 = sc0 sc1 nt) -> profiles/r04/merl_cold_policy.txt
# library: the tree of commit 40cea38 ("experiment: MERL gathers of cold rows...": the kernel reads DJB_MERL_COLD_POLICY / DJB_MERL_HOT_ROWS), built as the shipped one
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
O=gpurun_out/merl_cold_policy.txt; : > $O
run() { # policy rows workload
  A=""; case $3 in merl_eval_*) A="--n 250000000";; esac
  DJB_MERL_COLD_POLICY=$1 DJB_MERL_HOT_ROWS=$2 timeout 300 python bench.py --workload $3 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary $A 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('policy $1 hot_rows %-5s %-24s %8.3f ms/step  frac %.3f' % ('$2', '$3', d['ms_per_step'], d['roofline']['frac']))" >> $O
}
DJB_MERL_COLD_POLICY=1 timeout 900 python -m pytest tests/test_gpu_verification.py tests/test_gpu_golden.py tests/test_gpu_parity.py -m gpu -q -x > gpurun_out/pytest_cold_policy.log 2>&1; tail -2 gpurun_out/pytest_cold_policy.log >> $O
for rep in 1 2; do
  run 0 1699 merl_eval
  for p in 1 2 3; do run $p 1699 merl_eval; done
done
for r in 970 1213 1456 1941 2400 3000; do run 1 $r merl_eval; done
for p in 0 1; do for w in merl_eval_uniform_bins merl_eval_coherent; do run $p 1699 $w; done; done
cat $O
