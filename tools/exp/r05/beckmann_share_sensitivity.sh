#!/bin/bash
# round 5: how the contract sampler's time depends on the error model's constants -- timing-only builds with epsv(u) scaled by 0.5 / 0.25
# (bands and per-sample bound shrink with it; NOT shippable as such: the constants are what the directed attack was run against)
# builds: make -C dj_brdf_amd/csrc BUILD=build_cts05 OUT=../../gpurun_variants/libdjb_cts05.so EXTRA=-DDJB_EXP_CTS_SCALE=0.5f   (025 likewise)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
O=gpurun_out/beckmann_share_sensitivity.txt; : > $O
for rep in 1 2; do for v in shipped cts05 cts025; do
  lib=gpurun_variants/libdjb_$v.so; [ $v = shipped ] && lib=dj_brdf_amd/lib/libdjb_hip.so
  DJB_LIB_PATH=$lib timeout 300 python bench.py --workload beckmann_sample_contract --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('%-8s %8.3f ms/step  frac %.3f' % ('$v', d['ms_per_step'], d['roofline']['frac']))" >> $O
done; done
cat $O
