#!/bin/bash
# round 4: the contract-mode Beckmann sampler, first measurement
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
PYTHONPATH=. timeout 1200 python tools/contract_sample_probe.py ${1:-2.7e8} > gpurun_out/contract_sample.txt 2>&1; cat gpurun_out/contract_sample.txt
