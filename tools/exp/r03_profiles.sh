#!/bin/bash
# round 3: the rocprofv3 passes behind profiles/r03 (kernel stats, FETCH/WRITE/SQ/L2 counters) + the instruction mix of Beckmann sample
cd "${GRAFT_REPO_ROOT:-/root/repo}"
bash tools/profile_bench.sh > gpurun_out/profile_bench.log 2>&1
bash tools/instmix.sh beckmann_sample > gpurun_out/instmix_beckmann_sample.txt 2>&1
bash tools/instmix.sh ggx_eval_pdf_contract > gpurun_out/instmix_ggx_contract.txt 2>&1
tail -3 gpurun_out/profile_bench.log; head -20 gpurun_out/instmix_beckmann_sample.txt
