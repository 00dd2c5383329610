#!/bin/bash
# lib_ab.sh "<variant names>" "<workloads>": the shipped library against gpurun_variants/libdjb_<name>.so on the same box, ms per step of
# each bench workload (plain bench line, no CPU baseline / secondary legs) -> gpurun_out/lib_ab_<tag>.txt
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; OUT=gpurun_out/lib_ab_${TAG:-x}.txt; : > $OUT
for w in $2; do
  for lib in shipped $1; do
    if [ $lib != shipped ]; then export DJB_LIB_PATH=$PWD/gpurun_variants/libdjb_$lib.so; else unset DJB_LIB_PATH; fi
    timeout 600 python bench.py --workload $w --steps ${STEPS:-10} --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d.get('roofline') or {}
print('%-34s %-10s ms_per_step %9.4f frac %s' % ('$w', '$lib', d['ms_per_step'], r.get('frac')))" >> $OUT
  done
done
cat $OUT
