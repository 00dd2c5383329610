#!/bin/bash
# round 3, GPU call 23: GGX qf2 with the four addition forms merged (operands selected, one double division, guarded): parity + rates
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; O=gpurun_out/r03; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q -k "sample or evalp_is or golden or facade or scalar or queries or guarded" > $O/ggx_sample_tests.txt 2>&1; echo "rc=$?" >> $O/ggx_sample_tests.txt; tail -3 $O/ggx_sample_tests.txt
PYTHONPATH=. timeout 300 python tools/sample_rates.py 2>/dev/null
