#!/bin/bash
# round 4: tier 1 follows the reference's theta_h snap (h.z > 0.99999) instead of handing those pairs to tier 2
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_verification.py tests/test_gpu_golden.py tests/test_gpu_parity.py -m gpu -q -x -k "merl or guard" > gpurun_out/pytest_merl_snap.log 2>&1; tail -4 gpurun_out/pytest_merl_snap.log
O=gpurun_out/merl_snap.txt; : > $O
for w in merl_eval merl_eval_uniform_bins merl_eval_coherent merl_eval merl_eval_uniform_bins merl_eval_coherent; do
  A=""; case $w in merl_eval_*) A="--n 250000000";; esac
  timeout 300 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-secondary $A 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('%-24s %8.3f ms/step  frac %.3f' % ('$w', d['ms_per_step'], d['roofline']['frac']))" >> $O
done
cat $O
PYTHONPATH=. timeout 900 python tools/merl_guard_attack.py > gpurun_out/merl_guard_attack_snap.txt 2>&1; tail -3 gpurun_out/merl_guard_attack_snap.txt
PYTHONPATH=. timeout 600 python tools/calibrate_merl_guard.py > gpurun_out/merl_guard_calibration_snap.txt 2>&1; tail -25 gpurun_out/merl_guard_calibration_snap.txt
