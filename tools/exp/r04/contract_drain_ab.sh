#!/bin/bash
# round 4: contract-mode sgd / Beckmann eval with tier 2 drained in the kernel (k_ct_drain_v4, DJB_CT_DRAIN=1, the default) against the worklist +
# variant: the tree of the commit "experiment: contract-mode sgd / Beckmann with tier 2 drained in the kernel"
# fix-up kernel form (DJB_CT_DRAIN=0) -> profiles/r04/contract_drain_ab.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
O=gpurun_out/contract_drain_ab.txt; : > $O
for rep in 1 2; do for d in 0 1; do
  echo "== DJB_CT_DRAIN=$d" >> $O
  DJB_CT_DRAIN=$d timeout 300 python bench.py --workload sgd_eval_contract --steps 10 --warmup 10 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('sgd_eval_contract %8.3f ms per 1e8' % d['ms_per_step'])" >> $O
  DJB_CT_DRAIN=$d PYTHONPATH=. timeout 300 python tools/contract_beckmann_share.py 2>&1 | grep isotropic >> $O
done; done
cat $O
