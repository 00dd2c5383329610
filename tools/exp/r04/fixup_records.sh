#!/bin/bash
# round 4: the fix-up kernels walk worklist RECORDS (numbered through the shards) instead of list slots -> profiles/r04/fixup_records.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_verification.py tests/test_gpu_golden.py tests/test_gpu_parity.py tests/test_gpu_contract.py -m gpu -q -x > gpurun_out/pytest_fixup.log 2>&1; tail -3 gpurun_out/pytest_fixup.log
cd /tmp && export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for w in merl_eval merl_eval_uniform_bins; do
  A="--workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-secondary"; case $w in merl_eval_*) A="$A --n 250000000";; esac
  rm -rf gpurun_out/fixup_$w; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fixup_$w -- python bench.py $A > gpurun_out/fixup_$w.json 2>/dev/null
  python - $w <<'PY'
import csv, glob, sys
for f in glob.glob(f"gpurun_out/fixup_{sys.argv[1]}/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "k_merl" in r["Name"]: print(sys.argv[1], r["Name"][28:50], "calls", r["Calls"], "avg ms %.3f" % (float(r["AverageNs"]) * 1e-6), "min %.3f" % (float(r["MinNs"]) * 1e-6))
PY
  tail -1 gpurun_out/fixup_$w.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w', 'ms_per_step %.3f frac %.3f' % (d['ms_per_step'], d['roofline']['frac']))"
done
