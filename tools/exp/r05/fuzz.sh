#!/bin/bash
# round 5: randomised differential run of the final tree (the Beckmann sampler's last trip changed this round) + the sampler stress
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
for seed in ${SEEDS:-9601 9602 9603}; do PYTHONPATH=. timeout 1500 python tests/fuzz_parity.py 8 8e6 $seed 2>&1 | tail -4; done > gpurun_out/fuzz_r05.txt
PYTHONPATH=. timeout 1200 python tools/sampler_stress.py 2>&1 | tail -12 >> gpurun_out/fuzz_r05.txt
cat gpurun_out/fuzz_r05.txt
