#!/bin/bash
# round 4: the rocprofv3 passes behind profiles/r04 (kernel stats, FETCH / WRITE / SQ / L2 counters for every workload) + the
# instruction mixes behind profiles/valu_*.json
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
bash tools/profile_bench.sh > gpurun_out/profile_bench.log 2>&1
for w in merl_eval beckmann_sample beckmann_sample_contract ggx_eval_pdf ggx_eval_pdf_contract; do
  bash tools/instmix.sh $w > gpurun_out/instmix_$w.txt 2>&1
  n=1e9; case $w in ggx*) n=1e8;; esac
  python tools/valu_report.py $w $n
done
tail -3 gpurun_out/profile_bench.log
