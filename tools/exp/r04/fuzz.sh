#!/bin/bash
# round 4: randomised differential run of the final tree (the MERL tier changed this round) + sampler stress
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
for seed in ${SEEDS:-9501 9502 9503}; do PYTHONPATH=. timeout 1500 python tests/fuzz_parity.py 8 8e6 $seed 2>&1 | tail -4; done > gpurun_out/fuzz_r04.txt
cat gpurun_out/fuzz_r04.txt
