#!/bin/bash
# round 4: MERL tier 2 drained inside the tier-1 kernel (per-wave LDS queue, one dense wave of exact fp64 evaluations per 64 queued pairs)
# instead of a second kernel behind a worklist in HBM -> profiles/r04/merl_fused_tiers.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_verification.py tests/test_gpu_golden.py tests/test_gpu_parity.py tests/test_gpu_scalar_path.py -m gpu -q -x > gpurun_out/pytest_fused.log 2>&1; tail -3 gpurun_out/pytest_fused.log
O=gpurun_out/merl_fused_tiers.txt; : > $O
for rep in 1 2; do for w in merl_eval merl_eval_uniform_bins merl_eval_coherent; do
  A=""; case $w in merl_eval_*) A="--n 250000000";; esac
  timeout 300 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-secondary $A 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('%-24s %8.3f ms/step  frac %.3f' % ('$w', d['ms_per_step'], d['roofline']['frac']))" >> $O
done; done
cat $O
