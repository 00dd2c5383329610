#!/bin/bash
# round 3, GPU call 16/17: Beckmann sample two-path kernel: deferred share by cause (counting build), grid sweep, parity
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; O=gpurun_out/r03; mkdir -p $O
B="python bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-secondary"
DJB_LIB_PATH=gpurun_variants/libdjb_rc.so timeout 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-secondary --workload beckmann_sample --n 250000000 > $O/bk_rc.json 2> $O/bk_rc.err
grep djb_exp $O/bk_rc.err | tail -1
for g in 1024 2048 4096 8192 16384 976563; do
  DJB_SAMPLE_GRID_ENV=$g DJB_LIB_PATH=gpurun_variants/libdjb_ex.so timeout 300 $B --workload beckmann_sample --n 250000000 > $O/bk_grid_$g.json 2>/dev/null
  python -c "import json;print($g, '%.3f' % json.loads(open('$O/bk_grid_$g.json').read().strip().splitlines()[-1])['ms_per_step'])"
done
timeout 300 $B --workload beckmann_sample > $O/bk3_full.json 2>/dev/null
python -c "import json;print('full 1e9', '%.3f' % json.loads(open('$O/bk3_full.json').read().strip().splitlines()[-1])['ms_per_step'])"
timeout 1500 python -m pytest tests -m gpu -x -q -k "sample or histogram or evalp_is" > $O/sample_tests.txt 2>&1; echo "rc=$?" >> $O/sample_tests.txt; tail -3 $O/sample_tests.txt
