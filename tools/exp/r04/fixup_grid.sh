#!/bin/bash
# round 4: grid of k_merl_fixup (-DDJB_MERL_FIXUP_GRID=n variants in gpurun_variants/libdjb_fg<n>.so; shipped 2048)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for v in base fg512 fg1024 fg4096 fg8192 fg16384 base; do
  lib=gpurun_variants/libdjb_$v.so; [ $v = base ] && lib=dj_brdf_amd/lib/libdjb_hip.so
  rm -rf gpurun_out/fg; DJB_LIB_PATH=$lib rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fg -- python bench.py --workload merl_eval --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > /dev/null 2>&1
  python - $v <<'PY'
import csv, glob, sys
for f in glob.glob("gpurun_out/fg/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "k_merl_fixup" in r["Name"]: print(sys.argv[1], "k_merl_fixup avg ms %.3f min %.3f" % (float(r["AverageNs"]) * 1e-6, float(r["MinNs"]) * 1e-6))
PY
done
