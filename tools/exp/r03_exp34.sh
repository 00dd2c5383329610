#!/bin/bash
# round 3, GPU call 34: two-instruction near_f32_midpoint: full GPU suite, guarded self-test, rates
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; O=gpurun_out/r03; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/gpu_suite.txt 2>&1; echo "rc=$?" >> $O/gpu_suite.txt; tail -3 $O/gpu_suite.txt
PYTHONPATH=. timeout 600 python tools/kind_rates.py > $O/kind_rates_exact.txt 2>&1; grep -iE "eval" $O/kind_rates_exact.txt | head -12
PYTHONPATH=. timeout 600 python tools/sample_rates.py > $O/sample_rates.txt 2>&1; grep -iE "sample|evalp" $O/sample_rates.txt | head
for i in 1 2; do timeout 600 python bench.py --workload beckmann_sample --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('beckmann_sample ms', r['ms_per_step'], 'frac', r['roofline']['frac'])"; done
