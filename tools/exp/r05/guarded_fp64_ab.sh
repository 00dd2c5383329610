#!/bin/bash
# round 5 (last session): A/B on one box, alternating, of the guarded fp64 forms at three more float(<double>) sites --
# unpolarized Fresnel's float(sqrt(n^2 + c^2 - 1)) (sqrt_to_f32), abc's float(A / pow(...)) x 3 and sgd's float(ndf) (div_to_f32).
# base = gpurun_variants/libdjb_base.so (the tree before the change, built by `make OUT=...` from `git archive HEAD~`), new = the shipped library
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
{
for rep in 1 2 3; do
  for v in base new; do
    if [ $v = base ]; then export DJB_LIB_PATH=$PWD/gpurun_variants/libdjb_base.so; else unset DJB_LIB_PATH; fi
    echo "== $v (pass $rep)"
    PYTHONPATH=. timeout 600 python tools/kind_rates.py 2>&1 | grep -i "unpol\|abc\|sgd\|^ggx  "
  done
done
} > gpurun_out/guarded_fp64_ab.txt 2>&1
cat gpurun_out/guarded_fp64_ab.txt
