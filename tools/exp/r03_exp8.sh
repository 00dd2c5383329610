#!/bin/bash
# round 3, GPU call 8: contract mode incl. Beckmann (tests + rates), paced reaper loader rates
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; O=gpurun_out/r03; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_contract.py -x -q -s > $O/contract_tests2.txt 2>&1; echo "rc=$?" >> $O/contract_tests2.txt
timeout 300 python tools/kind_rates.py > $O/kind_rates_r03.txt 2>&1
DJB_KIND_RATES_CONTRACT=1 timeout 300 python tools/kind_rates.py > $O/kind_rates_contract.txt 2>&1
FIT_RATES_SPARSE_ONLY=1 timeout 600 python tools/fit_files_rates.py > $O/fit_files_rates_paced.txt 2>&1
tail -25 $O/contract_tests2.txt; grep -v amdgpu $O/kind_rates_contract.txt; grep -v amdgpu $O/fit_files_rates_paced.txt
