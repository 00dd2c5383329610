#!/bin/bash
# round 3, GPU call 36: sampler common path without its provably redundant flags: sampler tests, then A/B against gpurun_variants/libdjb_prev.so
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; O=gpurun_out/r03; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q -k "sampl or beckmann or verification or histogram" > $O/gpu_sampler.txt 2>&1; echo "rc=$?" >> $O/gpu_sampler.txt; tail -3 $O/gpu_sampler.txt
one() { timeout 600 python bench.py --workload $1 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2 ms %.3f' % r['ms_per_step'])"; }
for i in 1 2 3; do
  DJB_LIB_PATH=$R/gpurun_variants/libdjb_prev.so one beckmann_sample prev
  one beckmann_sample new
done
PYTHONPATH=. timeout 600 python tools/sample_rates.py 2>&1 | grep -iE "sample|evalp" | head
