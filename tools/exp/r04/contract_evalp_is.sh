#!/bin/bash
# round 4: evalp_is under the contract (exact direction, ct_is_tail for weight / pdf): tests + rates
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_contract.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/pytest_contract.log 2>&1; tail -6 gpurun_out/pytest_contract.log
{ PYTHONPATH=. timeout 600 python tools/sample_rates.py; DJB_SAMPLE_RATES_CONTRACT=1 PYTHONPATH=. timeout 600 python tools/sample_rates.py; } > gpurun_out/sample_rates.txt 2>&1; cat gpurun_out/sample_rates.txt
