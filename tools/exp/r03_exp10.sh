#!/bin/bash
# round 3, GPU call 10: MERL kernel with the sharded worklist: grid cap sweep (exp build), merl tests
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; O=gpurun_out/r03; mkdir -p $O
B="python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-secondary"
for cap in 8192 16384 32768 65536 131072 262144 1048576 100000000; do
  DJB_LIB_PATH=gpurun_variants/libdjb_exp.so DJB_MERL_GRID_CAP_ENV=$cap timeout 300 $B --workload merl_eval > $O/merlgrid2_$cap.json 2>$O/merlgrid2_$cap.err
done
timeout 900 python -m pytest tests/test_gpu_verification.py tests/test_gpu_golden.py tests/test_gpu_parity.py -x -q -k "merl" > $O/merl_tests2.txt 2>&1; echo "rc=$?" >> $O/merl_tests2.txt
python - <<'PY'
import json,glob
for cap in (8192,16384,32768,65536,131072,262144,1048576,100000000):
    try:
        r=json.loads(open(f"gpurun_out/r03/merlgrid2_{cap}.json").read().strip().splitlines()[-1]); print("cap %9d %8.3f ms frac %.3f" % (cap, r["ms_per_step"], r["roofline"]["frac"]))
    except Exception as e: print(cap, "ERR", e)
PY
tail -4 $O/merl_tests2.txt
