#!/bin/bash
# round 5: the contract-mode Beckmann sampler's last trip changed this round (the exit is shown from the Newton step) -- the directed attack
# and the self-test of round 4 again, on the final tree (Beckmann and GGX)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
PYTHONPATH=. timeout 2400 python tools/contract_sample_attack.py > gpurun_out/contract_sample_attack.txt 2>&1; tail -3 gpurun_out/contract_sample_attack.txt | cut -c1-200
PYTHONPATH=. timeout 2400 python tools/contract_sample_attack.py --ndf ggx > gpurun_out/contract_sample_attack_ggx.txt 2>&1; tail -2 gpurun_out/contract_sample_attack_ggx.txt | cut -c1-200
PYTHONPATH=. timeout 1200 python tools/contract_sample_probe.py 2.7e8 > gpurun_out/contract_sample.txt 2>&1; tail -8 gpurun_out/contract_sample.txt
