#!/bin/bash
# round 3, GPU call 4: derived guard constants (cost + attack), adaptive worklist on the uniform-bins leg, cndmask probe
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; O=gpurun_out/r03; mkdir -p $O
tools/bin/valu_cost_probe > $O/valu_cost2.txt 2>&1
B="python bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-secondary"
timeout 300 $B --workload merl_eval > $O/merl_newguard.json 2>$O/merl_newguard.err
timeout 300 $B --workload merl_eval_uniform_bins --n 250000000 > $O/uniform_adaptive.json 2>$O/uniform_adaptive.err
timeout 300 $B --workload merl_eval_coherent --n 250000000 > $O/coherent_newguard.json 2>$O/coherent_newguard.err
timeout 900 python tools/merl_guard_attack.py --m 262144 --iters 512 --rounds 3 > $O/merl_guard_attack.txt 2>$O/merl_guard_attack.err
timeout 600 python tools/merl_guard_attack.py --m 65536 --iters 256 --rounds 2 --guard 12 12 12 12 12 12 > $O/merl_guard_attack_r2consts.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_verification.py tests/test_gpu_golden.py -x -q -k "merl" > $O/merl_tests.txt 2>&1; echo "rc=$?" >> $O/merl_tests.txt
head -12 $O/valu_cost2.txt
python - <<'PY'
import json
for f in ("merl_newguard","uniform_adaptive","coherent_newguard"):
    try:
        r=json.loads(open(f"gpurun_out/r03/{f}.json").read().strip().splitlines()[-1])
        print("%-30s %8.3f ms  %8.2f G/s  frac %.3f" % (f, r["ms_per_step"], r["value"]/1e9, r["roofline"]["frac"] or 0))
    except Exception as e: print(f, "ERR", e)
PY
cat $O/merl_guard_attack.txt; tail -3 $O/merl_guard_attack.err; cat $O/merl_guard_attack_r2consts.txt | tail -16; tail -4 $O/merl_tests.txt
