#!/bin/bash
# round 3, GPU call 12: MERL table with five texels per 64-byte sector: the MERL tests, the whole suite, the three MERL legs
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; O=gpurun_out/r03; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gpu_suite4.txt 2>&1; echo "gpu suite rc=$?" >> $O/gpu_suite4.txt
B="python bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-secondary"
timeout 300 $B --workload merl_eval > $O/merl_sector.json 2>$O/merl_sector.err
timeout 300 $B --workload merl_eval > $O/merl_sector2.json 2>>$O/merl_sector.err
timeout 300 $B --workload merl_eval_uniform_bins --n 250000000 > $O/uniform_sector.json 2>>$O/merl_sector.err
timeout 300 $B --workload merl_eval_coherent --n 250000000 > $O/coherent_sector.json 2>>$O/merl_sector.err
tail -4 $O/gpu_suite4.txt
python - <<'PY'
import json
for f in ("merl_sector","merl_sector2","uniform_sector","coherent_sector"):
    try:
        r=json.loads(open(f"gpurun_out/r03/{f}.json").read().strip().splitlines()[-1]); print("%-20s %8.3f ms  %7.2f G/s  frac %.3f" % (f, r["ms_per_step"], r["value"]/1e9, r["roofline"]["frac"]))
    except Exception as e: print(f, "ERR", e)
PY
