#!/bin/bash
# round 3, GPU call 15: Beckmann sample as common path + deferred full path (djb_kernels_sample.hip): parity tests, rate
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; O=gpurun_out/r03; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q -k "sample or histogram or evalp_is or scalar or facade or golden" > $O/sample_tests.txt 2>&1; echo "rc=$?" >> $O/sample_tests.txt
B="python bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-secondary"
for rep in 1 2; do
  timeout 300 $B --workload beckmann_sample --n 250000000 > $O/bk2_$rep.json 2>$O/bk2.err
done
timeout 300 $B --workload beckmann_sample > $O/bk2_full.json 2>>$O/bk2.err
tail -5 $O/sample_tests.txt
python - <<'PY'
import json
for f in ("bk2_1","bk2_2","bk2_full"):
    try: print(f, "%.3f" % json.loads(open(f"gpurun_out/r03/{f}.json").read().strip().splitlines()[-1])["ms_per_step"])
    except Exception as e: print(f, "ERR", e)
PY
tail -3 $O/bk2.err
