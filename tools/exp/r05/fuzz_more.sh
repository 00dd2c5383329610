#!/bin/bash
# round 5: more seeds of the randomised differential run on the final tree
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
for seed in ${SEEDS:-9611 9612 9613 9614 9615 9616}; do PYTHONPATH=. timeout 1500 python tests/fuzz_parity.py 8 8e6 $seed 2>&1 | tail -2; done > gpurun_out/fuzz_r05_more.txt
cat gpurun_out/fuzz_r05_more.txt
