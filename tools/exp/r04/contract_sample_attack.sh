#!/bin/bash
# round 4: directed attack on the contract-mode sampler
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_contract.py -m gpu -x -q -k "attack or sample" > gpurun_out/pytest_contract_attack.log 2>&1; tail -4 gpurun_out/pytest_contract_attack.log
PYTHONPATH=. timeout 2400 python tools/contract_sample_attack.py > gpurun_out/contract_sample_attack.txt 2>&1; cat gpurun_out/contract_sample_attack.txt | cut -c1-200
