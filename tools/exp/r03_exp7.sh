#!/bin/bash
# round 3, GPU call 7: whole -m gpu suite on the refactored host code, reaper-thread loader rates, the default bench line
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; O=gpurun_out/r03; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gpu_suite3.txt 2>&1; echo "gpu suite rc=$?" >> $O/gpu_suite3.txt
FIT_RATES_SPARSE_ONLY=1 timeout 600 python tools/fit_files_rates.py > $O/fit_files_rates_reaper.txt 2>&1
timeout 1200 python bench.py > $O/bench_default.json 2>$O/bench_default.err
tail -5 $O/gpu_suite3.txt; grep -v amdgpu.ids $O/fit_files_rates_reaper.txt
python - <<'PY'
import json
r=json.loads(open("gpurun_out/r03/bench_default.json").read().strip().splitlines()[-1])
print({k:r[k] for k in ("metric","value","ms_per_step")}, r["roofline"]["frac"], r.get("cpu_baseline",{}).get("value"))
for k,v in r["secondary"].items():
    print(k, {a:(round(b,4) if isinstance(b,float) else b) for a,b in v.items() if a in ("value","ms_per_step","wall_ms","roofline_frac","max_rel_err_eval","max_rel_err_pdf","exact_tier_share","units_per_step","pipeline_ms")})
PY
tail -3 $O/bench_default.err
