#!/bin/bash
# round 3, GPU call 22: ABC in contract mode: tests, rate (tools/kind_rates.py)
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; O=gpurun_out/r03; mkdir -p $O
timeout 1800 python -m pytest tests/test_gpu_contract.py -x -q -s -k "abc" > $O/abc_contract_tests.txt 2>&1; echo "rc=$?" >> $O/abc_contract_tests.txt; grep -E "contract abc|passed|failed|Error|assert" $O/abc_contract_tests.txt | head -20
PYTHONPATH=. timeout 600 python tools/kind_rates.py > $O/kind_rates_exact.txt 2>&1; grep -iE "abc|sgd|ggx|beckmann" $O/kind_rates_exact.txt | head
DJB_KIND_RATES_CONTRACT=1 PYTHONPATH=. timeout 600 python tools/kind_rates.py > $O/kind_rates_contract.txt 2>&1; grep -iE "abc|sgd|ggx|beckmann" $O/kind_rates_contract.txt | head
