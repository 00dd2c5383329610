#!/bin/bash
# round 4: the tier-1 MERL estimate in fewer instructions (FMA contraction inside the tier, x rsq(x) instead of sqrt + rcp, one-reciprocal
# atan2, folded checks) x SLP vectorisation on / off for djb_kernels_merl.hip -> profiles/r04/merl_tier1_diet.txt
# (as run between commits 126344b and the tier-1 rewrite: old+slp = the library of 126344b; newslp / newnoslp / oldnoslp = the rewritten / old
#  djb_device_tables.inc with and without -fno-slp-vectorize for djb_kernels_merl.hip, built as variants into gpurun_variants/)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
O=gpurun_out/merl_tier1_diet.txt; : > $O
BASE=dj_brdf_amd/lib/libdjb_hip.so
run() { # name lib workload extra
  local line=$(DJB_LIB_PATH=$2 timeout 300 python bench.py --workload $3 --steps 10 --warmup 2 --no-cpu-baseline --no-secondary $4 2>/dev/null | tail -1)
  python - "$1" "$3" "$line" >> $O <<'PY'
import sys, json
try:
    d = json.loads(sys.argv[3]); print("%-10s %-24s %8.3f ms/step  launch %8.3f ms  frac %.3f" % (sys.argv[1], sys.argv[2], d["ms_per_step"], d["roofline"]["launch_ms"], d["roofline"]["frac"]))
except Exception as e:
    print(sys.argv[1], sys.argv[2], "FAILED", e, sys.argv[3][:200])
PY
}
for rep in 1 2; do
  run old+slp $BASE merl_eval
  for v in oldnoslp newslp newnoslp; do run $v gpurun_variants/libdjb_$v.so merl_eval; done
done
run old+slp $BASE merl_eval
for v in old+slp newnoslp; do
  lib=gpurun_variants/libdjb_$v.so; [ $v = old+slp ] && lib=$BASE
  run $v $lib merl_eval_uniform_bins "--n 250000000"
  run $v $lib merl_eval_coherent "--n 250000000"
done
cat $O
# correctness of the candidate: every MERL test of the GPU suite against the variant library
DJB_LIB_PATH=gpurun_variants/libdjb_newnoslp.so timeout 1200 python -m pytest tests/test_gpu_verification.py tests/test_gpu_golden.py tests/test_gpu_parity.py -m gpu -q -k "merl or guard" > gpurun_out/pytest_merl_newnoslp.log 2>&1
tail -5 gpurun_out/pytest_merl_newnoslp.log
DJB_LIB_PATH=gpurun_variants/libdjb_newnoslp.so bash tools/instmix.sh merl_eval _newnoslp > gpurun_out/instmix_merl_eval_newnoslp.txt 2>&1
python tools/valu_report.py merl_eval 1e9 "round 4 tier-1 diet, no SLP" _newnoslp | tee -a $O
DJB_LIB_PATH=gpurun_variants/libdjb_newnoslp.so timeout 900 python tools/merl_guard_attack.py > gpurun_out/merl_guard_attack_newnoslp.txt 2>&1; tail -15 gpurun_out/merl_guard_attack_newnoslp.txt
