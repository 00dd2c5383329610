#!/bin/bash
# round 6: utia::eval tier-1 forms A/B (DJB_UTIA_FORM: 0 = k_eval_utia_t1 angles first, 1 = k_utia_v2 lane-private fetch from estimated
# cells, 2 / 3 = k_utia_v2 wave-cooperative fetch through one / two LDS tiles), exact and under DJB_OPT_CONTRACT_1E5, per min-waves
# build (gpurun_variants/libdjb_wN.so = -DDJB_UTIA_V2_WAVES=N; the shipped library otherwise); parity first
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; OUT=gpurun_out/utia_v2.txt; : > $OUT
echo "== parity (shipped library)" >> $OUT
timeout 900 python -m pytest -x -q -m gpu tests/test_gpu_verification.py -k "utia" tests/test_gpu_parity.py::test_utia_eval tests/test_gpu_contract.py -k "utia" -s 2>&1 | grep -v "^$" | tail -25 >> $OUT
for lib in ${LIBS:-shipped w2 w3}; do
 for f in ${FORMS:-0 1 2 3}; do
  if [ $lib != shipped ]; then export DJB_LIB_PATH=$PWD/gpurun_variants/libdjb_$lib.so; [ $f = 0 ] && continue; else unset DJB_LIB_PATH; fi
  for w in utia_eval utia_eval_contract; do
    [ $f = 0 ] && [ $w = utia_eval_contract ] && continue
    DJB_UTIA_FORM=$f timeout 600 python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('lib $lib form $f $w ms_per_step %.3f frac %.4f' % (d['ms_per_step'], d['roofline']['frac']))" >> $OUT
  done
 done
done
cat $OUT
