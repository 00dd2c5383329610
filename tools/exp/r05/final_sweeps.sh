#!/bin/bash
# round 5: the directed searches and sweeps of earlier rounds again, on the final tree
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
PYTHONPATH=. timeout 1200 python tools/merl_guard_attack.py > gpurun_out/merl_guard_attack.txt 2>&1; tail -4 gpurun_out/merl_guard_attack.txt | cut -c1-220
PYTHONPATH=. timeout 1200 python tools/hostile_parity_sweep.py > gpurun_out/hostile_parity_sweep.txt 2>&1; tail -4 gpurun_out/hostile_parity_sweep.txt | cut -c1-220
PYTHONPATH=. timeout 1200 python tools/selftest_guarded.py > gpurun_out/selftest_guarded.txt 2>&1; tail -6 gpurun_out/selftest_guarded.txt | cut -c1-220
