#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
bash tools/exp/r04_exp5.sh 2.7e8 > /dev/null 2>&1; cat gpurun_out/contract_sample.txt
bash tools/exp/r04_exp6.sh > /dev/null 2>&1; grep -A1 "elliptic" gpurun_out/contract_sample_causes.txt
