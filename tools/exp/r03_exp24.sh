#!/bin/bash
# round 3, GPU call 24: guarded double quotient (div_to_f32) self-test: the suite's 2e9 inputs and a 1.28e11-input run
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; O=gpurun_out/r03; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "guarded" 2>&1 | tail -3
PYTHONPATH=. timeout 1500 python tools/selftest_guarded.py 32 > $O/selftest_guarded_div.txt 2>&1; tail -12 $O/selftest_guarded_div.txt
