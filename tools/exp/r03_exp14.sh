#!/bin/bash
# round 3, GPU call 14: Beckmann sample, bound of a "rare paths deferred" design: fixed 4 Newton trips and/or no w>=5 erfinv arm (timing only)
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; O=gpurun_out/r03; mkdir -p $O
B="python bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-secondary"
for rep in 1 2; do for v in ship na nf4 nf4na nf3na; do
  lib=$([ $v = ship ] && echo "" || echo gpurun_variants/libdjb_$v.so)
  DJB_LIB_PATH=$lib timeout 300 $B --workload beckmann_sample --n 250000000 > $O/defer_${v}_$rep.json 2>$O/defer_${v}.err
done; done
python - <<'PY'
import json
for v in ("ship","na","nf4","nf4na","nf3na"):
    print(v, ["%.3f" % json.loads(open(f"gpurun_out/r03/defer_{v}_{k}.json").read().strip().splitlines()[-1])["ms_per_step"] for k in (1,2)])
PY
