#!/bin/bash
# round 3, GPU call 17: Beckmann sample two-path kernel: grids that are whole multiples of the resident workgroups (256 CUs x 5)
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; O=gpurun_out/r03; mkdir -p $O
B="python bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-secondary"
for g in 1280 2560 5120 10240 20480 40960 81920 16384 32768; do
  DJB_SAMPLE_GRID_ENV=$g DJB_LIB_PATH=gpurun_variants/libdjb_ex.so timeout 300 $B --workload beckmann_sample --n 250000000 > $O/bk_grid_$g.json 2>/dev/null
  python -c "import json;print($g, '%.3f' % json.loads(open('$O/bk_grid_$g.json').read().strip().splitlines()[-1])['ms_per_step'])"
done
for g in 1280 5120 20480 81920; do
  DJB_SAMPLE_GRID_ENV=$g DJB_LIB_PATH=gpurun_variants/libdjb_ex.so timeout 300 $B --workload beckmann_sample > $O/bk_gridf_$g.json 2>/dev/null
  python -c "import json;print('1e9', $g, '%.3f' % json.loads(open('$O/bk_gridf_$g.json').read().strip().splitlines()[-1])['ms_per_step'])"
done
