#!/bin/bash
# round 3, GPU call 13: same-box A/B of the MERL table layouts: packed 12-byte texels (previous commit) vs five per 64-byte sector
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; O=gpurun_out/r03; mkdir -p $O
B="python bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-secondary"
for rep in 1 2 3; do for lib in gpurun_variants/libdjb_prev.so ""; do
  tag=$([ -z "$lib" ] && echo sector || echo packed)
  DJB_LIB_PATH=$lib timeout 300 $B --workload merl_eval > $O/lay_${tag}_$rep.json 2>/dev/null
  DJB_LIB_PATH=$lib timeout 300 $B --workload merl_eval_uniform_bins --n 250000000 > $O/layu_${tag}_$rep.json 2>/dev/null
done; done
python - <<'PY'
import json
for pre in ("lay","layu"):
    for tag in ("packed","sector"):
        print(pre, tag, ["%.3f" % json.loads(open(f"gpurun_out/r03/{pre}_{tag}_{k}.json").read().strip().splitlines()[-1])["ms_per_step"] for k in (1,2,3)])
PY
