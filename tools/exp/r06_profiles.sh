#!/bin/bash
# round 6: profile passes (kernel trace + stats, FETCH_SIZE, WRITE_SIZE, SQ_*) and instruction mixes for the given workloads
#   WORKLOADS="a b c" bash tools/exp/r06_profiles.sh        (run on the GPU box; then tools/summarize_profiles.py r06 here)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"
export WORKLOADS
bash tools/profile_bench.sh > gpurun_out/r06_profile_bench.log 2>&1
for w in $WORKLOADS; do
  case $w in fit_*|merl_fit) continue;; esac
  bash tools/instmix.sh $w > gpurun_out/instmix_$w.txt 2>&1
  n=$(python -c "import bench; print(bench.WORKLOADS['$w'][0])")      # instmix.sh runs the workload at its default size
  python tools/valu_report.py $w $n "round 6 (profiles/r06)" >> gpurun_out/r06_valu.log 2>&1
done
tail -30 gpurun_out/r06_valu.log
