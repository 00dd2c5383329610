#!/bin/bash
# round 5: contract-mode Beckmann sampler with two levels of deferral (declined samples -> the exact COMMON path as dense waves -> only
# its flagged samples -> sample_one): parity / contract tests, the selftest over lobes x families, timing
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
O=gpurun_out/beckmann_two_level.txt; : > $O
timeout 1500 python -m pytest tests/test_gpu_contract.py tests/test_gpu_parity.py -m gpu -x -q -k "sample or contract or beckmann" 2>&1 | tail -3 >> $O
for rep in 1 2 3; do for w in beckmann_sample beckmann_sample_contract; do
  timeout 300 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('%-26s %8.3f ms/step  frac %.3f' % ('$w', d['ms_per_step'], d['roofline']['frac']))" >> $O
done; done
PYTHONPATH=. timeout 1200 python tools/contract_sample_probe.py 1e8 2>&1 | tail -14 >> $O
cat $O
