#!/bin/bash
# round 5: utia::eval tier 1 with a wave-cooperative record fetch (k_eval_utia_coop, DJB_UTIA_COOP=1) against the lane-private form
# (the kernel lives in the commit "experiment: utia::eval tier 1 with a wave-cooperative record fetch"; it was removed again: slower)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
O=gpurun_out/utia_coop.txt; : > $O
echo "== parity with DJB_UTIA_COOP=1 (bit-identical to the oracle)" >> $O
DJB_UTIA_COOP=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_furnace.py tests/test_gpu_golden.py -m gpu -x -q -k "utia" 2>&1 | tail -3 >> $O
echo "== A/B, ms per 1e8 pairs (bench.py --workload utia_eval, 20 steps after 10 warm-up launches), alternating" >> $O
for rep in 1 2 3; do for v in 0 1; do
  DJB_UTIA_COOP=$v timeout 300 python bench.py --workload utia_eval --steps 20 --warmup 10 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('coop=$v  %8.3f ms/step  frac %.3f' % (d['ms_per_step'], d['roofline']['frac']))" >> $O
done; done
cat $O
