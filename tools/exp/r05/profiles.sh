#!/bin/bash
# round 5: the rocprofv3 passes behind profiles/r05 (kernel stats, FETCH / WRITE / SQ / L2 counters for every workload) and the
# instruction mixes behind profiles/valu_*.json -- all from the round's FINAL tree; afterwards, in the build container:
#   python tools/summarize_profiles.py r05
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; rm -f gpurun_out/valu_*.json
bash tools/profile_bench.sh > gpurun_out/profile_bench.log 2>&1
for w in merl_eval beckmann_sample beckmann_sample_contract ggx_eval_pdf ggx_eval_pdf_contract utia_eval; do
  bash tools/instmix.sh $w > gpurun_out/instmix_$w.txt 2>&1
  n=1e9; case $w in ggx*|utia*) n=1e8;; esac
  python tools/valu_report.py $w $n "round 5 (profiles/r05)"
done
tail -3 gpurun_out/profile_bench.log; ls gpurun_out/valu_*.json
