#!/bin/bash
# round 6, last session: same-box A/B of a library variant against the shipped one (run through gpurun).
#   make -C dj_brdf_amd/csrc BUILD=build_x OUT=../../gpurun_variants/libdjb_x.so EXTRA=-D...   (or a checkout of the previous commit)
#   VARIANTS="default x" WL="sgd_eval abc_evalp lean_evalp_pdf" bash tools/exp/r06/fast_tier_ab.sh
# DJB_SGD_FAST=0 in the environment turns the decided fast tier of sgd objects off (exact chains only) without a second library.
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { python bench.py --workload $1 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2 $1', round(d['ms_per_step'],4))"; }
for rep in 1 2; do for v in ${VARIANTS:-default}; do
  if [ $v = default ]; then unset DJB_LIB_PATH; else export DJB_LIB_PATH=$PWD/gpurun_variants/libdjb_$v.so; fi
  for w in ${WL:-sgd_eval abc_evalp}; do run $w $v; done
done; done
