#!/bin/bash
# round 3, GPU call 21: Beckmann sample two-path kernel: what fewer resident waves cost (LDS padding: 5 -> 4 -> 3 waves per SIMD)
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; O=gpurun_out/r03; mkdir -p $O
B="python bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-secondary"
for rep in 1 2; do for v in ship p16 p28 p36; do
  lib=$([ $v = ship ] && echo "" || echo gpurun_variants/libdjb_$v.so)
  DJB_LIB_PATH=$lib timeout 300 $B --workload beckmann_sample > $O/occ_${v}_$rep.json 2>/dev/null
  python -c "import json;print('$v', '%.3f' % json.loads(open('$O/occ_${v}_$rep.json').read().strip().splitlines()[-1])['ms_per_step'])"
done; done
