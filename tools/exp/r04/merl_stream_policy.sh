#!/bin/bash
# round 4: cache-policy bits of the MERL kernel's 36 B/pair streams (DJB_STREAM_LOAD_POLICY / DJB_STREAM_STORE_POLICY variants,
# djb_worklist.hpp) -> profiles/r04/merl_stream_policy.txt
# (variants: `make -C dj_brdf_amd/csrc BUILD=build_s1 OUT=../../gpurun_variants/libdjb_s1.so EXTRA=-DDJB_STREAM_STORE_POLICY=1` etc.;
#  sN = store policy N, lN = load policy N, l1s1 = both 1)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
O=gpurun_out/merl_stream_policy.txt; : > $O
run() { # name lib workload extra
  local lib=$2
  local line=$(DJB_LIB_PATH=$lib timeout 300 python bench.py --workload $3 --steps 10 --warmup 2 --no-cpu-baseline --no-secondary $4 2>/dev/null | tail -1)
  python - "$1" "$3" "$line" >> $O <<'PY'
import sys, json
try:
    d = json.loads(sys.argv[3]); print("%-8s %-24s %8.3f ms/step  launch %8.3f ms  frac %.3f" % (sys.argv[1], sys.argv[2], d["ms_per_step"], d["roofline"]["launch_ms"], d["roofline"]["frac"]))
except Exception as e:
    print(sys.argv[1], sys.argv[2], "FAILED", e, sys.argv[3][:200])
PY
}
BASE=dj_brdf_amd/lib/libdjb_hip.so
for rep in 1 2; do
  run base $BASE merl_eval
  for v in s1 s2 s3 s4 l3 l4 l1s1; do run $v gpurun_variants/libdjb_$v.so merl_eval; done
done
run base $BASE merl_eval
for v in base s1 s2 l1s1; do
  lib=gpurun_variants/libdjb_$v.so; [ $v = base ] && lib=$BASE
  run $v $lib merl_eval_uniform_bins "--n 250000000"
  run $v $lib merl_eval_coherent "--n 250000000"
done
cat $O
