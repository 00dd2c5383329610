#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_contract.py -m gpu -x -q > gpurun_out/pytest_contract.log 2>&1; tail -6 gpurun_out/pytest_contract.log
PYTHONPATH=. timeout 1200 python tools/contract_sample_probe.py 2.7e8 > gpurun_out/contract_sample.txt 2>&1; tail -8 gpurun_out/contract_sample.txt
